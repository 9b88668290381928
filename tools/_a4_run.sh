cd /root/repo
export A4_IMPLS=13,15
A4_ONLY="NS fwd,cube,XL fwd,base fwd,packed" timeout 300 python tools/a4_check.py 2>&1 | grep -v "amdgpu.ids"
A4_ONLY="NS dgrad,NS wgrad,base" timeout 300 python tools/a4_check.py --forms 2>&1 | grep -v "amdgpu.ids"
