cd /root/repo
export A4_IMPLS=13,15 A4_NOLIB=1
A4_ONLY="ffn1 gelu,packed large ffn1" timeout 300 python tools/a4_check.py 2>&1 | grep -v "amdgpu.ids" | sed 's/   library.*TF)//'
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -x -k "gelu or forward_forms" 2>&1 | tail -2
