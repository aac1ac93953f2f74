cd $GRAFT_REPO_ROOT
python tools/diag/count_diff.py 2>&1 | grep -v "missing\|base row" | tail -9
python -m pytest tests -m gpu -q -x -k "bench_configuration or view_list or full_size or end_to_end or fuse_and_clean or map_stages or lookahead or owner_sharded_map or config5_two or tiled_raster" > gpurun_out/r04_i_tests.log 2>&1; tail -4 gpurun_out/r04_i_tests.log
