cd $GRAFT_REPO_ROOT
IFX_OPTS="host_entry_async=1" python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | grep -a "passed\|failed\|^FAILED" | head -12
