# Direct 1x1 kernel: workgroup target sweep at ResNet50 / b256 / S = 16 sizes (env BT_DIRECT_WGS overrides launch_direct's policy).
for sh in r50l2ds r50l3ds r50l4ds r50l3c1 r50l4c1 r50l2c1 r50l4c3 r50l3c3 r50l1c3; do for t in 0 256 1024 4096; do
    BT_DIRECT_WGS=$t python tools/microbench.py --shape $sh --S 16 --B 256 --iters 5 --sigma 2>&1 | tail -1 | sed "s/^/wgs=$t /" | cut -c1-75
done; done
