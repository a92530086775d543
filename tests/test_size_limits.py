"""CPU: the 32-bit offset guards of the fused paths (no device needed)."""
from deepviewagg_amd import fused_bilinear


def test_fused_bilinear_size_limits():
    ok = fused_bilinear.size_limits_ok
    R, C = 32 * 64 * 128, 64
    assert ok(1 << 25, R, 1 << 20, C)                    # the headline size (z_a is 4 GiB there: per-tile descriptors)
    assert ok((1 << 26) - 32, R, (1 << 21) - 1, 32)       # the largest scene of 32-view points
    assert not ok(1 << 26, R, 1 << 21, 32)                # 64-byte chain rows: V x 64 reaches 2^32
    assert not ok(1 << 20, (1 << 25), 1 << 15, 64)        # map rows Y [R, C] bf16 beyond 4 GiB
    assert not ok(1 << 20, R, 1 << 25, 64)                # per-point rows beyond 4 GiB
    assert ok(1 << 22, R, 1 << 17, 128) and ok(1 << 22, R, 1 << 17, 256)
