# timing ablations / experiments of the quad-row kernel: builds of liblcf.so with -DQ4_ABLATE=<bits>
# or -DQ4_PRIO=<bits> linked as tools/ubench/liblcf_q4<tag>.so, each through tools/bench_q4.py
set -e -o pipefail
export Q4_ONLY=${Q4_ONLY:-q4x4} Q4_NOERR=1
for v in base "$@"; do
  if [ $v = base ]; then unset LCF_LIB; else export LCF_LIB=tools/ubench/liblcf_q4$v.so; fi
  echo "== $v"
  timeout -k 10 120 python3 tools/bench_q4.py > gpurun_out/q4ab_$v.log 2>&1
  grep -v amdgpu.ids gpurun_out/q4ab_$v.log
done
