cd $GRAFT_REPO_ROOT
T=r04_zz
( time python -m pytest tests -m gpu -q ) > gpurun_out/${T}_tests.log 2>&1; tail -4 gpurun_out/${T}_tests.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
