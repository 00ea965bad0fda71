cd /tmp && export TMPDIR=/tmp
for v in ${VARIANTS:-fewbase}; do
  export MNF_LIB_PATH=$GRAFT_REPO_ROOT/tools/bin/libmnf_$v.so
  rm -rf $GRAFT_REPO_ROOT/gpurun_out/few_prof
  timeout 300 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/few_prof -o few -- python3 $GRAFT_REPO_ROOT/tools/time_rnvp_few.py child > /dev/null 2>&1
  echo "== $v"; python3 $GRAFT_REPO_ROOT/tools/few_summary.py $GRAFT_REPO_ROOT/gpurun_out/few_prof/few_kernel_trace.csv | cat
done
