export FX_BGZF_GROUP=0
python tools/bgzf_decode_probe.py 3.0 build/libfxgpu_regmap.so build/libfxgpu_regmap2.so build/libfxgpu_regmap.so build/libfxgpu_regmap2.so 2>&1 | tail -4
