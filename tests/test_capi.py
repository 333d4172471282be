"""The C-ABI library: loads, exports every symbol include/cfhip.h declares, the ctypes table matches
the header, and argument validation returns errors (no compute without a GPU)."""
import ctypes
import os
import re

import pytest

import cflearn_amd
from cflearn_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, "include", "cfhip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(cfhip_\w+)\s*\(", text)))


def test_header_symbols_exported():
    lib = _lib.load()
    names = _declared()
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), f"libcfhip.so does not export {n}"


def test_ctypes_table_matches_header():
    assert sorted(_lib.SIGNATURES) == _declared()


def test_version_and_error_string():
    lib = _lib.load()
    assert lib.cfhip_version() == 100
    # invalid arguments are rejected before any launch, with a message
    rc = lib.cfhip_gemm_bf16(None, None, None, None, None, None, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0, 1, None, 0, None, 0, None)
    assert rc == -1
    assert b"null" in lib.cfhip_last_error()
    rc = lib.cfhip_layernorm_fwd(1, 0, 1, 1, 1, None, None, 4, 6, 8, 8, 1e-6, None)
    assert rc == -1 and b"multiple of 4" in lib.cfhip_last_error()
    # (long sequences are no longer an error: the chunked kernels take them) head_dim outside the supported set is
    rc = lib.cfhip_attn_fwd_dh(16, 16, 16, 16, None, None, 1, 1, 300, 300, 200, 256, 256, 256, 256, 256, 256, 0, 0, 0,
                               0.125, 0, None)
    assert rc == -1 and b"head_dim" in lib.cfhip_last_error()
    with pytest.raises(RuntimeError, match="head_dim"):
        _lib.check(rc, "attn_fwd")


def test_workspace_queries():
    lib = _lib.load()
    assert lib.cfhip_colsum_workspace(12608, 768) >= 768 * 4
    assert lib.cfhip_layernorm_bwd_workspace(12608, 768) >= 2 * 768 * 4


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "libcfhip.so"))
    with pytest.raises(RuntimeError, match="no fallback"):
        _lib.load()


def test_ops_refuse_cpu_tensors():
    import torch

    a = torch.zeros(8, 8, dtype=torch.bfloat16)
    with pytest.raises(RuntimeError, match="HIP device"):
        cflearn_amd.ops.gemm(a, a)


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "carefree-learn_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                src = open(os.path.join(dp, f)).read()
                assert "vit_oracle" not in src and "refharness" not in src and "import oracle" not in src, f


def test_conv3x3_host_side_checks_and_split_heuristic():
    """No launch happens: argument validation and the split-K decision of the implicit-GEMM convolution."""
    lib = _lib.load()
    # one K-step = 32 channels of a tap
    rc = lib.cfhip_conv3x3_nhwc_bf16(16, 16, None, 16, 1, 8, 8, 48, 64, None, 0, None)
    assert rc == -1 and b"multiple of 32" in lib.cfhip_last_error()
    rc = lib.cfhip_conv3x3_nhwc_bf16(16, 16, None, 16, 1, 8, 8, 64, 60, None, 0, None)
    assert rc == -1 and b"multiple of 8" in lib.cfhip_last_error()
    rc = lib.cfhip_conv3x3_wgrad_nhwc_bf16(16, 16, 16, 0, None, 0, 1, 1, 8, 64, 64, 1, None, 0, None)
    assert rc == -1 and b"bad image shape" in lib.cfhip_last_error()
    rc = lib.cfhip_conv3x3_pack_filters(16, 16, 8, 12, 0, None)
    assert rc == -1
    # zoo UNet at 64^2 x 8: level 0 (32768 pixels x 320) fills the chip -> no split, no workspace; the deeper levels
    # (8192 x 640: 160 tiles, 2048 x 1280: 80, 512 x 1280: 40) split and ask for split * pixels * Cout fp32
    assert lib.cfhip_conv3x3_workspace(8, 64, 64, 320, 320) == 0
    for (hw, cin, cout) in ((32, 640, 640), (16, 1280, 1280), (8, 1280, 1280)):
        nbytes = lib.cfhip_conv3x3_workspace(8, hw, hw, cin, cout)
        pixels = 8 * hw * hw
        assert nbytes > 0 and nbytes % (pixels * cout * 4) == 0
        split = nbytes // (pixels * cout * 4)
        assert 2 <= split <= (9 * cin // 64) // 8
    assert lib.cfhip_conv3x3_wgrad_workspace(320, 640, 4) == 4 * 640 * (9 * 320 + 1) * 4


def test_vectorcall_entry_module_is_the_same_library():
    """`_cfhip_fast` (csrc/gen_fastcall.py): METH_FASTCALL wrappers of the SAME loaded libcfhip.so, generated from
    `_lib.SIGNATURES` — every entry point whose arguments are plain numbers is served by it, the others (ctypes objects /
    strings at their call sites) stay on ctypes; wrong argument counts / types are Python errors raised BEFORE the library
    is entered; return codes and error strings are the library's."""
    if not os.path.isfile(_lib.FAST_PATH) or os.environ.get("CFHIP_FASTCALL", "1") == "0":
        pytest.skip("the vectorcall module is not built (build_lib.sh / __graft_entry__.build() builds it)")
    lib = _lib.load()
    assert type(lib).__name__ == "_FastLib" and _lib.fast_bound >= 75
    cdll = lib.__dict__["_cdll"]
    fast_names = [n for n in _lib.SIGNATURES if n in lib.__dict__]
    slow_names = [n for n in _lib.SIGNATURES if n not in lib.__dict__]
    assert "cfhip_gemm_bf16" in fast_names and "cfhip_adam_step_dev" in fast_names and "cfhip_attn_fwd_dh" in fast_names
    assert all(n.startswith("cfhip_comm_") or n in ("cfhip_version", "cfhip_last_error", "cfhip_set_option", "cfhip_gemm_kernel_name",
                                                     "cfhip_gemm_bf16_grouped_tn",
                                                     "cfhip_layernorm_bwd_partials", "cfhip_layernorm_bwd2") for n in slow_names), slow_names
    args = (None, None, None, None, None, None, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0, 1, None, 0, None, 0, None)
    assert lib.cfhip_gemm_bf16(*args) == cdll.cfhip_gemm_bf16(*args) == -1 and b"null" in lib.cfhip_last_error()
    assert lib.cfhip_colsum_workspace(1000, 64) == cdll.cfhip_colsum_workspace(1000, 64) > 0
    assert lib.cfhip_layernorm_bwd_workspace(512, 768) == cdll.cfhip_layernorm_bwd_workspace(512, 768)
    with pytest.raises(TypeError, match="23 arguments"):
        lib.cfhip_gemm_bf16(1, 2, 3)
    with pytest.raises(TypeError):
        lib.cfhip_colsum_workspace("a", 3)
    with pytest.raises((TypeError, OverflowError)):
        lib.cfhip_gemm_bf16(*(("x",) + args[1:]))


def test_kernels_spill_only_where_documented(tmp_path):
    """The code objects inside the built libcfhip.so: no kernel uses scratch memory except the three head_dim-64 long-sequence attention
    instantiations DESIGN.md §3.5 names (12-36 bytes; built without scratch they are 6-16 % slower, profiles/r05/attn_ndt4_scratch_ab.txt).
    Read from the kernels' metadata notes with the ROCm LLVM tools — a compiler flag or an edit that makes a kernel spill shows up here,
    on the CPU, before it costs anything on the GPU."""
    import re
    import shutil
    import subprocess

    llvm = "/opt/rocm/lib/llvm/bin"
    tools = [os.path.join(llvm, t) for t in ("llvm-objcopy", "clang-offload-bundler", "llvm-readelf")]
    if not all(os.path.exists(t) for t in tools) or not os.path.exists(_lib.LIB_PATH):
        pytest.skip("ROCm LLVM tools or the built library are missing")
    fat = tmp_path / "fat.bin"
    subprocess.run([tools[0], f"--dump-section=.hip_fatbin={fat}", _lib.LIB_PATH, str(tmp_path / "stripped.so")], check=True)
    data = fat.read_bytes()
    starts = [m.start() for m in re.finditer(re.escape(b"__CLANG_OFFLOAD_BUNDLE__"), data)]  # one bundle per translation unit
    assert len(starts) >= 10, len(starts)
    kernels, spilling = 0, {}
    for i, off in enumerate(starts):
        blob = tmp_path / f"bundle{i}.bin"
        blob.write_bytes(data[off:starts[i + 1] if i + 1 < len(starts) else len(data)])
        elf = tmp_path / f"code{i}.elf"
        subprocess.run([tools[1], "--unbundle", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--input={blob}",
                        f"--output={elf}"], check=True, capture_output=True)
        notes = subprocess.run([tools[2], "--notes", str(elf)], check=True, capture_output=True, text=True).stdout
        for name, scratch in re.findall(r"\.name:\s+(\S+)\n(?:.*\n)*?\s+\.private_segment_fixed_size:\s+(\d+)", notes):
            kernels += 1
            if int(scratch) > 0:
                spilling[name] = int(scratch)
    shutil.rmtree(tmp_path, ignore_errors=True)
    assert kernels > 400, kernels
    allowed = ("attn_fwd2_kernelILb1ELi4ELb0E", "attn_fwd2_kernelILb0ELi4ELb0E", "attn_bwd_dq2_kernelILb1ELi4E")
    unexpected = {k: v for k, v in spilling.items() if not any(a in k for a in allowed)}
    assert not unexpected, unexpected
    assert all(v <= 64 for v in spilling.values()), spilling
