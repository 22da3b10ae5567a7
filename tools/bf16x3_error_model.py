"""Host model: a K-term fp32 dot product evaluated (a) as an fp32 fma chain (what v_mfma_f32_32x32x2_f32 does per output),
(b) with both operands split into three bf16 terms and the six largest cross products accumulated in fp32 (what six
v_mfma_f32_32x32x16_bf16 per K chunk would do: a0b0 + a0b1 + a1b0 + a1b1 + a0b2 + a2b0), (c) with two-term splits and three
products (a0b0 + a0b1 + a1b0) — each against the float64 result, on activations / weights shaped like the generator's layers
(ReLU-ed inputs, weights ~ N(0, 1/(9 Cin))).  Prints max and rms error relative to the output scale.  No GPU needed.

    python tools/bf16x3_error_model.py
"""
import numpy as np


def bf16(x):
    """round-to-nearest-even to bfloat16, returned as float32"""
    u = x.astype(np.float32).view(np.uint32).astype(np.uint64)
    r = ((u >> 16) & 1) + 0x7FFF
    return (((u + r) >> 16) << 16).astype(np.uint32).view(np.float32)


def split(x, n):
    parts, rest = [], x.astype(np.float32)
    for _ in range(n):
        p = bf16(rest)
        parts.append(p)
        rest = (rest - p).astype(np.float32)
    return parts


def chain32(terms):
    """sum over the last axis in fp32, k-ordered (one rounding per addition)"""
    acc = np.zeros(terms.shape[:-1], np.float32)
    for k in range(terms.shape[-1]):
        acc = (acc + terms[..., k]).astype(np.float32)
    return acc


def main():
    rng = np.random.default_rng(0)
    print("%-22s %12s %12s | %12s %12s | %12s %12s" % ("K (= 9 Cin)", "fp32 max", "rms", "3x bf16 max", "rms", "2x bf16 max", "rms"))
    for cin in (32, 128, 512, 1024):
        K, n = 9 * cin, 4096
        a = np.maximum(rng.standard_normal((n, K)), 0).astype(np.float32)
        b = (rng.standard_normal((n, K)) / np.sqrt(K)).astype(np.float32)
        ref = (a.astype(np.float64) * b.astype(np.float64)).sum(-1)
        scale = np.abs(ref).max()
        # (a) fp32 fma chain: the product is exact inside the fma, one rounding per step
        acc = np.zeros(n, np.float64)
        for k in range(K):
            acc = (acc + a[:, k].astype(np.float64) * b[:, k].astype(np.float64)).astype(np.float32).astype(np.float64)
        e32 = acc - ref
        out = []
        for nsplit, pairs in ((3, ((0, 0), (0, 1), (1, 0), (1, 1), (0, 2), (2, 0))), (2, ((0, 0), (0, 1), (1, 0)))):
            sa, sb = split(a, nsplit), split(b, nsplit)
            acc = np.zeros(n, np.float64)
            for k in range(K):              # per k: the cross products (exact in fp32), smallest first, into the fp32 accumulator
                for (i, j) in reversed(pairs):
                    acc = (acc + sa[i][:, k].astype(np.float64) * sb[j][:, k].astype(np.float64)).astype(np.float32).astype(np.float64)
            out.append(acc - ref)
        print("%-22s %12.2e %12.2e | %12.2e %12.2e | %12.2e %12.2e" % (
            "%d (Cin %d)" % (K, cin), np.abs(e32).max() / scale, np.sqrt((e32 ** 2).mean()) / scale,
            np.abs(out[0]).max() / scale, np.sqrt((out[0] ** 2).mean()) / scale,
            np.abs(out[1]).max() / scale, np.sqrt((out[1] ** 2).mean()) / scale))


if __name__ == "__main__":
    main()
