set -e
for sh in conv1 layer1 layer2 layer3 layer4 ds4 l4s; do for m in 1 0; do timeout -k 5 60 python tools/microbench.py --shape $sh --sigma --kl --mode $m $( [ $sh = conv1 ] && echo --shared ); done; done
