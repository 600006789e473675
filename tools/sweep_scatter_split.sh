# channel-group split of the backward scatter: cost model (default) vs the halving rule of rounds 1-2 vs forced group counts
set -e
mkdir -p gpurun_out/r03d
O=gpurun_out/r03d/scatter_split.txt
: > $O
for shape in "N=2 C=512 Co=512 dil=4 pad=4" "N=8 C=256 Co=256 dil=2 pad=2" "N=8 C=128 Co=128 dil=1 pad=1" "N=4 C=512 Co=512 H=60 W=80 dil=4 pad=4" "N=4 C=256 Co=256 H=60 W=80 dil=2 pad=2"; do
 for knob in "DDCN_SCATTER_SPLIT_MODEL=0" "DDCN_SCATTER_SPLIT_MODEL=1" "DDCN_SCATTER_GPD=3" "DDCN_SCATTER_GPD=5" "DDCN_SCATTER_GPD=6" "DDCN_SCATTER_GPD=7" "DDCN_SCATTER_GPD=12"; do
   echo "== $shape $knob" >> $O
   env $knob timeout -k 10 120 python tools/stage_bench.py $shape col2im_coord >> $O 2>&1
 done
done
cat $O
