for v in "" emb_NOMFMA emb_NOEPI emb_NODMA emb_NOMFMA_NOEPI emb_NOMFMA_NODMA; do
  if [ -n "$v" ]; then export DPOT_HIP_LIB=$PWD/dpot_amd/lib/variants/libdpot_hip_$v.so; fi
  echo -n "$v: "; python scripts/small_kernels_bench.py 2>&1 | grep "embed_fwd"
done
