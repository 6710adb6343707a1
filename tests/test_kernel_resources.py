"""The register budgets the headline kernels are tuned for, read from the BUILT library's code objects (no GPU needed).

The wave-per-stream kernels run four workgroups of four waves per compute unit: 512 vector registers per SIMD / 4 waves = 128, and the
two analysis halves are tuned to stay clear of that edge (<= 120).  DESIGN section 3 records that the 48 kHz / 10 ms front half lands
at 117 registers / 0 spilled only in the translation unit it is compiled in today and at 128 + 47 spilled (7 % slower) in a unit of
its own, from almost the same IR: a compiler update or an unrelated edit can move it silently.  This test is the tripwire."""
import importlib
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
pkg = importlib.import_module("lc3-codec_amd")

# kernel (mangled-name fragment of the 48 kHz / 10 ms view) -> (most vector registers, why)
BUDGET = {
    "lc3_enc_front_kernelI13lc3_cfg_48k10E": (120, "analysis front half: four waves per SIMD with room to spare"),
    "lc3_enc_back_kernelI13lc3_cfg_48k10E": (120, "analysis back half: four waves per SIMD with room to spare"),
    "lc3_decode_kernelI13lc3_cfg_48k10E": (128, "synthesis: four waves per SIMD"),
    "lc3_sns_vq_kerneliPfPiii": (256, "vector quantiser, lane per frame: one or two waves per SIMD"),
    "lc3_pack_pc_kerneliPKiPh": (128, "packer pair, lane per frame: a producer and a consumer wave per SIMD beside another kernel's waves"),
    "lc3_parse_pc_kernelI13lc3_cfg_48k10E": (168, "parser pair, lane per frame: three waves per SIMD"),
}

# Vector registers a kernel may spill: none (the synthesis kernel, AT its 128 registers, carried a 64-bit lane offset in scratch across the
# frame loop for a while in round 5; requesting the state ahead of the frame loop's own first loads removed it).
SPILLS_ALLOWED = {}


@pytest.fixture(scope="module")
def rows():
    import kernel_resources as KR

    if not os.path.exists(os.path.join(KR.LLVM_BIN, "llvm-objdump")):
        pytest.skip("no llvm-objdump / llvm-readelf under " + KR.LLVM_BIN)
    return KR.from_library(pkg.build_native())


def test_headline_kernels_keep_their_register_budgets(rows):
    for frag, (most, why) in BUDGET.items():
        hit = [r for r in rows if frag in r["name"]]
        assert len(hit) == 1, (frag, [r["name"] for r in hit])
        r = hit[0]
        assert r.get("vgpr_spill_count", 0) <= SPILLS_ALLOWED.get(frag, 0), f"{r['name']}: {r['vgpr_spill_count']} vector registers spilled ({why})"
        assert r["vgpr_count"] <= most, f"{r['name']}: {r['vgpr_count']} vector registers, budget {most} ({why})"


def test_wave_per_stream_kernels_fit_four_workgroups_per_compute_unit(rows):
    # 160 KB of LDS per compute unit: four workgroups of a wave-per-stream kernel ("workgroup slots")
    for frag in ("lc3_enc_front_kernelI13lc3_cfg_48k10E", "lc3_enc_back_kernelI13lc3_cfg_48k10E", "lc3_decode_kernelI13lc3_cfg_48k10E"):
        r = [r for r in rows if frag in r["name"]][0]
        assert 4 * r["group_segment_fixed_size"] <= 160 * 1024, (r["name"], r["group_segment_fixed_size"])


def test_every_compile_time_view_has_its_kernels(rows):
    # the multi-unit library carries the three encoder and the pair / synthesis decoder kernels of all twelve views
    views = ["48k10", "48k75", "32k10", "16k10", "44k10", "24k10", "32k75", "24k75", "16k75", "44k75", "8k10", "8k75"]
    names = [r["name"] for r in rows]
    for v in views:
        for k in ("lc3_enc_front_kernel", "lc3_enc_back_kernel", "lc3_decode_kernel", "lc3_parse_pc_kernel"):
            assert any(k in n and "lc3_cfg_" + v in n for n in names), (k, v)
