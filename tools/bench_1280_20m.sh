for o in "--opt raster_tiles=1" "--opt raster_tiles=0" "--opt raster_tiles=0 --opt view_list=0"; do
  python bench.py --gpus 1 --steps 100 --warmup 20 --no-cpu-baseline --extras-frames 0 --res 1280x960 --surfels 20000000 $o 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['roofline']['kernels']
print('$o', d['value'], d['ms_per_frame_gpu'], d['view_list'], {n:round(v['avg_ms']*1000,1) for n,v in k.items() if v['avg_ms']>0.02})"
done
