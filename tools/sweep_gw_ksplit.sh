set -e
mkdir -p gpurun_out/r03
O=gpurun_out/r03/gw_ksplit.txt
: > $O
for shape in "N=2 C=512 Co=512 dil=4 pad=4" "N=8 C=256 Co=256 dil=2 pad=2" "N=8 C=128 Co=128 dil=1 pad=1" "N=8 C=512 Co=512 dil=4 pad=4"; do
 echo "== $shape (default)" >> $O
 timeout -k 10 120 python tools/stage_bench.py $shape gemm_gweight 2>&1 | grep gemm >> $O
 for s in 1 2 3 4 5 6 7 8 10 12 14 16; do
   echo "== $shape DDCN_GEMM_GW_KSPLIT=$s" >> $O
   DDCN_GEMM_GW_KSPLIT=$s timeout -k 10 120 python tools/stage_bench.py $shape gemm_gweight 2>&1 | grep gemm >> $O
 done
done
cat $O
