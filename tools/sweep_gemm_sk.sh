# G2 / G3 stage kernels: flat grid (DDCN_GEMM_SK=0), persistent kernel forced (=1) and the default choice, on model layer shapes
set -e
mkdir -p gpurun_out/r03
O=gpurun_out/r03/gemm_sk.txt
: > $O
for shape in "N=2 C=512 Co=512 dil=4 pad=4" "N=8 C=256 Co=256 dil=2 pad=2" "N=8 C=128 Co=128 dil=1 pad=1" "N=8 C=512 Co=512 dil=4 pad=4" "N=4 C=128 Co=128 H=60 W=80 dil=1 pad=1" "N=4 C=256 Co=256 H=60 W=80 dil=2 pad=2" "N=4 C=512 Co=512 H=60 W=80 dil=4 pad=4"; do
 for sk in 0 1 default; do
   echo "== $shape DDCN_GEMM_SK=$sk" >> $O
   if [ $sk = default ]; then timeout -k 10 120 python tools/stage_bench.py $shape gemm_gcol gemm_gweight 2>&1 | grep gemm >> $O
   else DDCN_GEMM_SK=$sk timeout -k 10 120 python tools/stage_bench.py $shape gemm_gcol gemm_gweight 2>&1 | grep gemm >> $O; fi
 done
done
cat $O
