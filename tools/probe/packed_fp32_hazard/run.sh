#!/usr/bin/env bash
# Reproducer of the packed-fp32 hazard (round 6): builds the library AS IT WAS BUILT BEFORE (packed fp32 instructions allowed) next to the
# shipped one, builds the co-tenant micro-kernels, and runs the victim beside each co-tenant with both libraries.
#   bash tools/probe/packed_fp32_hazard/run.sh  ->  gpurun_out/packed_fp32_hazard.txt
cd ${GRAFT_REPO_ROOT:-$(dirname $0)/../../..}
D=gpurun_out/hazard; mkdir -p $D/o
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
for f in oa-dg_amd/csrc/*.hip; do
  $HIPCC --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -munsafe-fp-atomics -Iinclude -Ioa-dg_amd/csrc -Wno-unused-function \
         -c $f -o $D/o/$(basename ${f%.hip}).o 2>/dev/null &
done; wait
$HIPCC --offload-arch=gfx950 -shared -fPIC $D/o/*.o -lz -o $D/liboadg_hip_slp_vectorized.so
$HIPCC --offload-arch=gfx950 -O3 -shared -fPIC tools/probe/packed_fp32_hazard/mfma_tenant.hip -o $D/libtenant.so
$HIPCC --offload-arch=gfx950 -O3 -fno-slp-vectorize -shared -fPIC tools/probe/packed_fp32_hazard/micro_victims.hip -o $D/libvictims.so
$HIPCC --offload-arch=gfx950 -O3 -ffp-contract=off -shared -fPIC tools/probe/packed_fp32_hazard/micro_victim9.hip -o $D/libvictim9.so
export TENANT_LIB=$PWD/$D/libtenant.so VICTIM_LIB=$PWD/$D/libvictims.so VICTIM9_LIB=$PWD/$D/libvictim9.so
{
for lib in $PWD/$D/liboadg_hip_slp_vectorized.so shipped; do
  for t in none valu mfma16 mfma32 conv128; do
    if [ $lib = shipped ]; then unset OADG_HIP_LIB; else export OADG_HIP_LIB=$lib; fi
    TENANT=$t timeout 120 python tools/probe/packed_fp32_hazard/victim.py 2>&1 | grep "^library"
  done
done
echo; echo "instruction forms (micro_victims.hip, 2048 x 256 lanes x 4000 dependent iterations per launch):"
timeout 200 python tools/probe/packed_fp32_hazard/micro.py 2>&1 | grep "^co-tenant"
} | tee gpurun_out/packed_fp32_hazard.txt
