# A/B library variant built in the build container (travels to the GPU box with the snapshot): rebuilds the given units with extra flags and
# links them with the in-tree objects.   bash tools/build_variant_lib.sh <name> "<flags>" [unit ...]   (default unit: gcrnn_fused_seq32p)
#   -> gated_gcrnns_amd/lib/variants/<name>.so ; run with GCRNN_LIBPATH=gated_gcrnns_amd/lib/variants/<name>.so python bench.py ...
R=$(cd "$(dirname "$0")/.." && pwd)
C=$R/gated_gcrnns_amd/csrc; L=$R/gated_gcrnns_amd/lib; N=$1; FL="$2"; shift; shift
U=${@:-gcrnn_fused_seq32p}
mkdir -p $L/variants /tmp/varobj_$N
EX=""
for u in $U; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -Wno-unused-result $FL -c $C/$u.hip -o /tmp/varobj_$N/$u.hip.o &
  EX="$EX -e /$u.hip.o"
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $L/variants/$N.so /tmp/varobj_$N/*.o $(ls $L/*.o | grep -v $EX)
ls -la $L/variants/$N.so
