export FX_BGZF_GROUP=0
python tools/bgzf_decode_probe.py 3.0 pyfastx_amd/csrc/libfxgpu.so 2>&1 | tail -1
FX_BGZF_REPLAY=1 python tools/bgzf_decode_probe.py 3.0 pyfastx_amd/csrc/libfxgpu.so 2>&1 | tail -1
