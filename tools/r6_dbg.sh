R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6n; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
ok=0; bad=0
for i in $(seq 1 25); do
  timeout 300 python3 -m pytest $R/tests -m gpu -x -q -k "gathers_partial or model_options or another_thread" > $O/loop_$i.log 2>&1
  if [ $? = 0 ]; then ok=$((ok+1)); rm -f $O/loop_$i.log; else bad=$((bad+1)); fi
done
echo "toggle loop: ok=$ok bad=$bad"
timeout 1500 python3 -m pytest $R/tests -m gpu -x -q > $O/full.log 2>&1; echo "full rc=$?"; tail -2 $O/full.log
