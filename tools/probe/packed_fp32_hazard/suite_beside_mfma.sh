#!/usr/bin/env bash
# The whole GPU parity suite while a second process keeps every SIMD busy with bare matrix instructions: which tests fail with
# the library built as before (SLP vectorizer on), and do all pass with the shipped one?
#   bash tools/probe/packed_fp32_hazard/suite_beside_mfma.sh -> gpurun_out/suite_beside_mfma.txt   (run.sh first: it builds the libraries)
cd ${GRAFT_REPO_ROOT:-$(dirname $0)/../../..}
D=gpurun_out/hazard
[ -f $D/liboadg_hip_slp_vectorized.so ] || bash tools/probe/packed_fp32_hazard/run.sh > /dev/null 2>&1
export TENANT_LIB=$PWD/$D/libtenant.so
python tools/probe/packed_fp32_hazard/tenant.py mfma16 > /dev/null 2>&1 &
TP=$!
sleep 20
{
echo "== shipped library beside the mfma16 co-tenant"
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | grep -E "^FAILED|^ERROR|passed|failed" | tail -40
if [ "$OLD" = 1 ]; then
echo "== library built as before (packed fp32 on) beside the mfma16 co-tenant"
OADG_HIP_LIB=$PWD/$D/liboadg_hip_slp_vectorized.so timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | grep -E "^FAILED|^ERROR|passed|failed" | tail -60
fi
} | tee gpurun_out/suite_beside_mfma.txt
kill $TP
