cd /tmp; export TMPDIR=/tmp; mkdir -p $GRAFT_REPO_ROOT/gpurun_out/r06
R=$GRAFT_REPO_ROOT
rm -rf /tmp/kt; timeout 300 rocprofv3 --kernel-trace --memory-copy-trace -d /tmp/kt -o kt -- $R/tools/scratch/slab_delay_ring 12 1 30 16 4 > /tmp/kt.log 2>&1
tail -3 /tmp/kt.log
db=$(find /tmp/kt -name "*.db" | head -1)
python - <<PY
import sqlite3
c=sqlite3.connect("$db")
print([r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")][:60])
PY
python $R/tools/rocpd_timeline.py $db 0 1 | head -3
n=$(python -c "
import sqlite3
c=sqlite3.connect('$db'); print(c.execute('select count(*) from kernels').fetchone()[0])")
echo total kernels $n
# the last run = overlap + wire: its backward is at the end
python $R/tools/rocpd_timeline.py $db $((n-60)) 60 > $R/gpurun_out/r06/delay_ring_timeline_tail.txt
cat $R/gpurun_out/r06/delay_ring_timeline_tail.txt
