# Diagnostic library with in-kernel phase stamps, built where hipcc is (the build container: the .so travels to the GPU box with the snapshot):
# the two wide units with -DGCRNN_SEQ_STAMPS [+ extra flags], linked with the in-tree objects of the other units.
#   bash tools/build_stamp_lib.sh [name] ["extra flags"]   ->  gated_gcrnns_amd/lib/variants/<name>.so   (use with GCRNN_STAMP_LIB=... python3 tools/seq32_stamps.py [native])
R=$(cd "$(dirname "$0")/.." && pwd)
C=$R/gated_gcrnns_amd/csrc; L=$R/gated_gcrnns_amd/lib; N=${1:-stamps}; FL="-DGCRNN_SEQ_STAMPS $2"
mkdir -p $L/variants /tmp/stampobj_$N
for u in gcrnn_fused_seq32 gcrnn_fused_seq32p; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -Wno-unused-result $FL -c $C/$u.hip -o /tmp/stampobj_$N/$u.hip.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $L/variants/$N.so /tmp/stampobj_$N/*.o $(ls $L/*.o | grep -v "gcrnn_fused_seq32p\?\.hip\.o")
ls -la $L/variants/$N.so
