cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -q -x -k "host_entry_returning" 2>&1 | grep -a "^E  \|assert\|passed\|failed" | head -12 | cut -c1-300
