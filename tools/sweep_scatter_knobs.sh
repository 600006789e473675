set -e
mkdir -p gpurun_out/r03
O=gpurun_out/r03/scatter_knobs.txt
: > $O
for shape in "N=2 C=512 Co=512 dil=4 pad=4" "N=8 C=256 Co=256 dil=2 pad=2"; do
 for knob in "" "DDCN_SCATTER_SC=2" "DDCN_TILE_MARGIN=4" "DDCN_SCATTER_ADAPTIVE=0" "DDCN_SCATTER_CG=32" "DDCN_SCATTER_CG=128"; do
   echo "== $shape offstd=2 $knob" >> $O
   env $knob timeout -k 10 120 python tools/stage_bench.py $shape offstd=2 col2im_coord 2>&1 | grep col2im >> $O
 done
done
cat $O
