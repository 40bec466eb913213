for v in base p0 p4 p8; do
  for sh in r50l3c3 r50l3c1 r50l1c3 r50l2ds; do
    BT_LIB_PATH=$GRAFT_REPO_ROOT/bayesian_torch_amd/libbtorch_hip_$v.so python tools/microbench.py --shape $sh --S 8 --B 256 --iters 10 --sigma 2>&1 | tail -1 | sed "s/^/$v /" | cut -c1-150
  done
done
