cd $GRAFT_REPO_ROOT
f=0
for n in $(seq 1 25); do
python -m pytest tests -m gpu -q -x -k "rccl_world_of_one_in_library" > /tmp/t_$n.log 2>&1 || { f=$((f+1)); grep -a "^E  \|assert" /tmp/t_$n.log | head -5; }
done
echo "isolated: failures $f of 25"
# with a test before it in the same process (state left behind in the allocator)
f=0
for n in $(seq 1 6); do
python -m pytest tests -m gpu -q -x -k "owner_sharded or config5_two_streams or rccl_world" > /tmp/u_$n.log 2>&1 || { f=$((f+1)); grep -a "^FAILED\|^E  " /tmp/u_$n.log | head -5; }
done
echo "group: failures $f of 6"
