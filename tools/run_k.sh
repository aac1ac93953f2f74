cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -q -x -k "camera_contexts_on_the_enqueue_path" 2>&1 | grep -a "^E  \|passed\|failed" | head -10 | cut -c1-300
