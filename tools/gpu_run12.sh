cd $GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm version\|^Hostname\|^Librccl" > /tmp/full.log
grep -n "def test_arena_reducer_hooks" -A 60 /tmp/full.log | grep -A40 "^[0-9]*[:-]>" | head -60
tail -5 /tmp/full.log
