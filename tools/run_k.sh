cd $GRAFT_REPO_ROOT
IFX_SEG_TRACE=1 python bench.py --steps 100 --warmup 20 --no-cpu-baseline --extras-frames 0 2>&1 >/dev/null | grep "seg call" | head -8
