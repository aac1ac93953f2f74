cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -q -x -k "three_streams" 2>&1 | grep -a "passed\|failed\|^FAILED\|^E  " | head
