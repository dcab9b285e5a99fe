IFA_PROMPT_LENS=40,64,128,256,512 IFA_AB_OPTION=prefill_mid IFA_BIG_MINS=0,1 timeout 600 python tools/bench_prompt_lens.py 2>&1 | grep -v amdgpu.ids
echo "== no GLU split (bit-identity check against the large-tile kernel)"
IFA_MID_NO_GLU_SPLIT=1 IFA_PROMPT_LENS=64,128,200 IFA_AB_OPTION=prefill_mid IFA_BIG_MINS=0,1 timeout 600 python tools/bench_prompt_lens.py 2>&1 | grep -v amdgpu.ids
