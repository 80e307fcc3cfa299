"""The C-ABI libraries load and export every symbol include/*.h declares (no compute calls)."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared(header, prefix):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(%s[a-z0-9_]+)\s*\(" % prefix, text)))


def test_host_library_exports_rt_host_h(rt):
    L = rt._ffi.host_lib()
    names = declared("rt_host.h", "rth_")
    assert len(names) >= 40
    for n in names:
        assert hasattr(L, n), n
    assert sorted(rt._ffi.HOST_SYMBOLS) == names


def test_hip_library_exports_rt_api_h(rt):
    path = os.path.join(rt._ffi.LIB_DIR, "librtiow_hip.so")
    assert os.path.exists(path), "HIP extension not built: run __graft_entry__.build()"
    L = ctypes.CDLL(path)   # links libamdhip64; needs no GPU
    names = declared("rt_api.h", "rt_")
    assert len(names) == 32      # the 14 of round 1 + rt_trace_first_hit_noalloc + rt_device_* (4) + rt_stream_* (3) + rt_comm_* (4) + rt_group_* (5) + rt_last_kernel_stats
    for n in names:
        assert hasattr(L, n), n
    assert sorted(rt._ffi.HIP_SYMBOLS) == names


def test_struct_sizes_match_the_headers(rt, tmp_path):
    """ctypes mirrors vs sizeof() as the C compiler sees the headers."""
    import subprocess
    src = tmp_path / "sz.c"
    src.write_text('#include <stdio.h>\n#include "rt_host.h"\nint main(){printf("%zu %zu %zu %zu %zu %zu %zu %zu %zu %zu\\n",'
                   'sizeof(rt_scene_desc),sizeof(rt_sampler_desc),sizeof(rt_camera),sizeof(rt_render_params),sizeof(rt_stats),'
                   'sizeof(rtg_node),sizeof(rtg_graph),sizeof(rth_user_config),sizeof(rth_scene_assets),sizeof(rt_scene_info));return 0;}\n')
    exe = tmp_path / "sz"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    got = [int(x) for x in subprocess.check_output([str(exe)]).split()]
    f = rt._ffi
    want = [ctypes.sizeof(t) for t in (f.rt_scene_desc, f.rt_sampler_desc, f.rt_camera, f.rt_render_params, f.rt_stats,
                                       f.rtg_node, f.rtg_graph, f.rth_user_config, f.rth_scene_assets, f.rt_scene_info)]
    assert got == want


def test_no_gpu_means_a_loud_error_not_a_fallback(rt):
    """Without a device rt_create must fail with RT_E_NO_DEVICE; the package has no CPU render path."""
    L = rt._ffi.hip_lib()
    if L.rt_device_count() > 0:
        import pytest
        pytest.skip("a GPU is present")
    import pytest
    with pytest.raises(rt.RtError) as e:
        rt.backend.HipBackend(0)
    assert e.value.status == -6
    src = open(os.path.join(ROOT, "raytracing-in-one-weekend_amd", "backend.py")).read()
    assert "oracle" not in src.replace("CPU oracle", "")


def test_the_c_host_compiles_against_the_headers_as_they_are():
    """integration/c_host.c carries _Static_asserts on every struct's size and field offsets (what integration/gpu_ffi.rs mirrors):
    a header change that forgets them — or a c_host binary left over from before it — must not go unnoticed.  Syntax and
    assertions only (no link: the libraries are not needed for this)."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run(["gcc", "-std=c11", "-fsyntax-only", "-Wall", "-I", os.path.join(root, "include"), os.path.join(root, "integration", "c_host.c")],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
