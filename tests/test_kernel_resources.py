"""CPU: what the gfx950 code objects inside libparakeet_slam.so say about every kernel -- read from the AMDGPU metadata notes
(no GPU needed).  No kernel that a default route can launch may use scratch memory: a spilled register costs a scratch
round trip that queues behind every row in flight (one in-order vector memory counter), which is how the first form of
k_step_regs lost 20 %."""
import os
import re
import struct
import subprocess
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
READELF = "/opt/rocm/lib/llvm/bin/llvm-readelf"

# kernels that still spill, each with the reason it is not on a default route
ALLOWED_SCRATCH = {
    "_ZN2pk11k_step_regsILb0EEEvNS_8RegsArgsE": 'k_step_regs<grid walk>: only with the option "cand_lists" = 0 (a scan whose candidate '
                                                 "lists overflow goes to the fall-back kernels instead)",
}


def code_object_kernels(so):
    data = open(so, "rb").read()
    out, idx = [], 0
    while True:
        i = data.find(b"\x7fELF", idx)
        if i < 0:
            break
        idx = i + 4
        if struct.unpack_from("<H", data, i + 18)[0] != 224:  # e_machine: EM_AMDGPU
            continue
        e_shoff = struct.unpack_from("<Q", data, i + 0x28)[0]
        e_shentsize, e_shnum = struct.unpack_from("<HH", data, i + 0x3A)
        with tempfile.NamedTemporaryFile(suffix=".co", delete=False) as f:
            f.write(data[i:i + e_shoff + e_shentsize * e_shnum])
        try:
            txt = subprocess.run([READELF, "--notes", f.name], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, check=True).stdout.decode()
        finally:
            os.unlink(f.name)
        cur = {}
        for line in txt.split("\n"):
            m = re.match(r"\s+-?\s*\.(\w+):\s+(.*)", line)
            if not m:
                continue
            k, v = m.group(1), m.group(2).strip()
            if k in ("private_segment_fixed_size", "sgpr_spill_count", "vgpr_spill_count", "vgpr_count", "symbol", "group_segment_fixed_size"):
                cur[k] = v
            if k == "wavefront_size":  # the last key of a kernel's record
                if "symbol" in cur:
                    out.append(cur)
                cur = {}
    return out


@pytest.fixture(scope="module")
def kernels():
    if not os.path.exists(READELF):
        pytest.skip("llvm-readelf not found")
    from parakeet_slam_amd import build

    return code_object_kernels(build.build(verbose=False))


def test_the_library_holds_the_kernels_of_the_path(kernels):
    names = " ".join(k["symbol"] for k in kernels)
    for needle in ("k_motion", "k_step_pub", "k_step_regs", "k_step_fused", "k_observe", "k_assoc_grid", "k_candidates", "k_cand_entries",
                   "k_scan_local", "k_ancestors", "k_summary"):
        assert needle in names, needle
    assert len(kernels) >= 50


def test_no_kernel_of_a_default_route_uses_scratch(kernels):
    bad = []
    for k in kernels:
        sym = k["symbol"].replace(".kd", "")
        scratch, spilled = int(k["private_segment_fixed_size"]), int(k["vgpr_spill_count"])
        if (scratch or spilled) and sym not in ALLOWED_SCRATCH:
            bad.append((sym, scratch, spilled))
    assert not bad, "kernels with scratch / spilled VGPRs: %r" % bad


def test_the_allow_list_is_not_stale(kernels):
    by = {k["symbol"].replace(".kd", ""): k for k in kernels}
    for sym in ALLOWED_SCRATCH:
        assert sym in by and int(by[sym]["private_segment_fixed_size"]) > 0, "%s no longer spills: drop it from ALLOWED_SCRATCH" % sym


def test_the_register_resident_kernels_keep_their_occupancy(kernels):
    """k_step_pub: 512 lanes x <= 256 VGPRs = two waves per SIMD, one workgroup per CU; k_step_regs<candidate lists>: 1 024 lanes
    x <= 128; k_step_fused: <= 128 (two 512-lane workgroups per CU)."""
    by = {k["symbol"].replace(".kd", ""): int(k["vgpr_count"]) for k in kernels}
    assert by["_ZN2pk10k_step_pubILi2ELi512EEEvNS_7PubArgsE"] <= 256
    assert by["_ZN2pk10k_step_pubILi1ELi256EEEvNS_7PubArgsE"] <= 168  # three 256-lane workgroups per CU
    assert by["_ZN2pk11k_step_regsILb1EEEvNS_8RegsArgsE"] <= 128
    for sym, v in by.items():
        if "k_step_fused" in sym:
            assert v <= 128, (sym, v)


def test_the_two_and_three_workgroups_per_cu_instances_of_the_two_pass_kernel_fit_their_registers(kernels):
    """k_step_pub_duo (round 6).  <NT, 1>: 512 lanes x <= 128 VGPRs = four waves per SIMD, TWO workgroups per CU; <NT, 2>: 256 lanes x
    <= 168 VGPRs = three waves per SIMD, THREE workgroups per CU -- by construction (one landmark at a time, one carried word per
    landmark), not by a cap the allocator answers with spills: no scratch, nothing spilled; and the static LDS beside 78 KB (50 KB) of
    dynamic LDS leaves room for the other workgroups (160 KB per CU)."""
    duo = [k for k in kernels if "k_step_pub_duo" in k["symbol"]]
    one = [k for k in duo if k["symbol"].replace(".kd", "").endswith("ELi1EEEvNS_7PubArgsE")]
    two = [k for k in duo if k["symbol"].replace(".kd", "").endswith("ELi2EEEvNS_7PubArgsE")]
    assert len(one) >= 2 and len(two) >= 2, [k["symbol"] for k in duo]
    for k in duo:
        assert int(k["private_segment_fixed_size"]) == 0 and int(k["vgpr_spill_count"]) == 0, k
    for k in one:
        assert int(k["vgpr_count"]) <= 128, k
        assert 2 * (int(k.get("group_segment_fixed_size", 0)) + 78 * 1024) <= 160 * 1024, k
    for k in two:
        assert int(k["vgpr_count"]) <= 168, k
        assert 3 * (int(k.get("group_segment_fixed_size", 0)) + 50 * 1024) <= 160 * 1024, k
