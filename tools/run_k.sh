cd $GRAFT_REPO_ROOT
for n in 1 2 3; do
python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | grep -a "passed\|failed\|^FAILED" | head -5
done
