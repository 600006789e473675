set -e
mkdir -p gpurun_out/r03
O=gpurun_out/r03/scatter_fx.txt
: > $O
for shape in "N=2 C=512 Co=512 dil=4 pad=4" "N=8 C=256 Co=256 dil=2 pad=2" "N=8 C=128 Co=128 dil=1 pad=1"; do
 for off in 0 2 8; do
  for fx in 1 0; do
   echo "== $shape offstd=$off DDCN_SCATTER_FX=$fx" >> $O
   DDCN_SCATTER_FX=$fx timeout -k 10 120 python tools/stage_bench.py $shape offstd=$off col2im_coord >> $O 2>&1
  done
 done
done
cat $O
