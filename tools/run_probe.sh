for l in pyfastx_amd/csrc/libfxgpu.so build/libfxgpu_FX_SC_WPE_6.so build/libfxgpu_FX_SC_WPE_4.so build/libfxgpu_FX_COMP_DEPTH_4.so pyfastx_amd/csrc/libfxgpu.so; do
  echo $l; FX_LIBFXGPU=/root/repo/$l python tools/fullindex_probe.py 3.0 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['one_read']['wall_ms'], d['one_read']['kernels_ms_avg'].get('k_scan_comp'), d['rows_equal'])"
done
