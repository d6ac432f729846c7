for reg in 1 0; do for ord in 0 1; do for pc in 1024 32 8; do
echo "== GCMF_HOST_REGISTER=$reg GCMF_HOST_ORDER=$ord GCMF_HOST_PIECE_MB=$pc"
GCMF_HOST_REGISTER=$reg GCMF_HOST_ORDER=$ord GCMF_HOST_PIECE_MB=$pc timeout 200 python tools/measure_host_batch.py 32 2>&1 | grep -v amdgpu.ids
done; done; done
