# timing ablations of the six-product kernel: builds of liblcf.so with -DBX_ABLATE=<bits> linked as
# tools/ubench/liblcf_bx<bits>.so, each through tools/bench_bx.py (results are wrong, timing only)
set -e -o pipefail
for v in base "$@"; do
  if [ $v = base ]; then unset LCF_LIB; else export LCF_LIB=tools/ubench/liblcf_bx$v.so; fi
  echo "== $v"
  timeout -k 10 120 python3 tools/bench_bx.py > gpurun_out/bxab_$v.log 2>&1
  grep -v -E "amdgpu.ids|err vs" gpurun_out/bxab_$v.log
done
