set -e
for sh in layer1 layer2 layer3 layer4 conv1 ds4; do timeout -k 5 60 python tools/microbench.py --shape $sh; done
timeout -k 5 60 python tools/microbench.py --shape layer1 --S 1 --B 4
timeout -k 5 60 python tools/microbench.py --shape layer1 --S 1 --B 4 --iters 200
timeout -k 5 60 python tools/microbench.py --shape layer1 --S 8 --B 128
