cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/c1prof -o c1 -- python $GRAFT_REPO_ROOT/scripts/c1_step_time.py > /dev/null 2>&1
cd $GRAFT_REPO_ROOT; for db in $(find /tmp/c1prof -name "*.db" | head -1); do python scripts/rocprof_summary.py $db | head -24 | cut -c1-170; done
