# same-box A/B of the late rounds: small whole proofs with tools/ab/libsumcheck_hip_prev.so and the current library (SC_SHAPE=c3|c2|gkr)
for rep in 1 2; do
  for L in tools/ab/libsumcheck_hip_prev.so sumcheck_amd/libsumcheck_hip.so; do
    echo -n "$L  "; SC_LIB_PATH=$PWD/$L timeout 200 python tools/small_proofs.py "$@" 2>&1 | grep "nv="
  done
done
