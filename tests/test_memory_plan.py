"""The HBM budget of one rank of the adaptive S-norm pipeline (pure arithmetic, no device): what is
resident, what a chunk of tests and a tile cost, which tile shape a given amount of free memory
allows, and the error raised before anything is allocated when the job cannot fit.  The reference
leaves this to its job grid (egs/voxceleb/v1/steps_be/eval_be_v1.sh:33-60: parts x parts processes)."""
import pytest

from hyperion_amd.backend import plan_snorm_memory

GB = 1e9
ROW = 848  # bytes of a packed row of 152 columns as fp16 pairs


def test_headline_workload_fits_in_one_tile():
    # 10 000 models x 1 000 000 tests, cohort 20 000, result kept: 40 GB result + 0.8 GB panel
    p = plan_snorm_memory(10_000, 20_000, 1_000_000, 300, ROW, 0.85 * 288 * GB, test_chunk=131072, x_dim=512)
    assert p["test_chunk"] == 131072 and p["enrol_block"] == 10_000
    assert abs(p["result"] - 10_000 * 1_000_000 * 4) < 1e6
    assert 0.8 * GB < p["resident"] < 0.9 * GB
    # the chunk: the [131072, 20000] panel (10.5 GB) + lists + packed rows; the tile: 4 x 5.2 GB
    assert 10.4 * GB < p["per_chunk"] < 11.2 * GB
    assert 20.9 * GB < p["per_tile"] < 21.2 * GB
    assert p["total"] == p["resident"] + p["result"] + p["per_chunk"] + p["per_tile"] < 0.85 * 288 * GB


def test_million_models_walk_the_enrolment_rows_in_blocks():
    # 1 000 000 models: the panel alone is 80 GB; streamed result, 32 768 tests per chunk
    p = plan_snorm_memory(1_000_000, 20_000, 125_000, 300, ROW, 0.85 * 288 * GB, test_chunk=32768,
                          materialise=False, x_dim=512)
    assert 80 * GB < p["resident"] < 84 * GB
    assert p["result"] == 0 and p["test_chunk"] == 32768
    assert p["enrol_block"] % 1024 == 0 and 1024 <= p["enrol_block"] < 1_000_000
    assert p["total"] <= 0.85 * 288 * GB
    # a cap on the tile alone gives smaller blocks, never a smaller chunk
    q = plan_snorm_memory(1_000_000, 20_000, 125_000, 300, ROW, 0.85 * 288 * GB, test_chunk=32768,
                          materialise=False, x_dim=512, tile_bytes=96 << 30)
    assert q["test_chunk"] == 32768 and q["enrol_block"] % 1024 == 0
    assert q["per_tile"] <= 96 << 30 and q["enrol_block"] <= p["enrol_block"]
    # after prepare_cohort the panel is held already: the same budget left beside it plans the same tiles
    r = plan_snorm_memory(1_000_000, 20_000, 125_000, 300, ROW, 0.85 * 288 * GB - p["resident"], test_chunk=32768,
                          materialise=False, x_dim=512, resident_allocated=True)
    assert (r["test_chunk"], r["enrol_block"]) == (p["test_chunk"], p["enrol_block"])


def test_short_memory_shrinks_the_chunk_then_refuses():
    # 60 GB: the 1M-model panel (80 GB) cannot be held at all
    with pytest.raises(MemoryError) as e:
        plan_snorm_memory(1_000_000, 20_000, 125_000, 300, ROW, 60 * GB, test_chunk=32768, materialise=False)
    assert "80.0 GB" in str(e.value) and "shard the enrolment set" in str(e.value)
    # 2 GB beside the resident part: the default chunk (2.7 GB) does not fit; the chunk is halved until a
    # 1024-row tile fits beside it
    res = plan_snorm_memory(1_000_000, 20_000, 125_000, 300, ROW, 200 * GB, test_chunk=32768,
                            materialise=False)["resident"]
    p = plan_snorm_memory(1_000_000, 20_000, 125_000, 300, ROW, res + 2 * GB, test_chunk=32768, materialise=False)
    assert p["test_chunk"] == 16384 and p["enrol_block"] >= 1024 and p["enrol_block"] % 1024 == 0
    assert p["total"] <= res + 2 * GB
    with pytest.raises(MemoryError):  # not even the smallest chunk
        plan_snorm_memory(1_000_000, 20_000, 125_000, 300, ROW, res + 0.3 * GB, test_chunk=32768, materialise=False)
    # a materialised 1M x 1M result (4 TB) is refused whatever the tile
    with pytest.raises(MemoryError):
        plan_snorm_memory(1_000_000, 20_000, 1_000_000, 300, ROW, 0.85 * 288 * GB)


def test_small_problems_are_one_tile():
    p = plan_snorm_memory(53, 400, 333, 50, ROW, 1 * GB, test_chunk=100)
    assert p["test_chunk"] == 100 and p["enrol_block"] == 53
    p = plan_snorm_memory(2500, 400, 700, 50, 4 * 16, 1 * GB, test_chunk=512, tile_bytes=24 * 512 * 1024)
    assert p["test_chunk"] == 512 and p["enrol_block"] == 1024
    p = plan_snorm_memory(5, 40, 0, 50, ROW, 1 * GB)  # no tests: nothing to tile
    assert p["enrol_block"] == 5
