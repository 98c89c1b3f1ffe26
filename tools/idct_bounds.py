"""Operand and output bounds of the reference's fixed-point IDCT network (src/wasm/mpeg1.c:1673-1740)
for ANY token stream: levels are clipped to [-2048, 2047] before the premultiplier (mpeg1.c:1546-1551),
the intra DC is dc << 8 with |dc| < 2^12 in valid streams.  Two facts k_recon relies on:
  1. every multiplication operand of the network stays below 2^23 in magnitude (v_mad_i32_i24 is exact),
  2. every output, after the final >> 8, stays inside int16 (the residual is kept as int16 pairs).
The network is piecewise linear up to its roundings, so the worst case over all inputs is bounded by the
sum over the 64 inputs of the absolute response to each input alone at full amplitude, plus the rounding
slack (each rounding moves a value by < 1 and gains along the network are < 4).
    python tools/idct_bounds.py
"""
import numpy as np

PM = [32, 44, 42, 38, 32, 25, 17, 9, 44, 62, 58, 52, 44, 35, 24, 12,
      42, 58, 55, 49, 42, 33, 23, 12, 38, 52, 49, 44, 38, 30, 20, 10,
      32, 44, 42, 38, 32, 25, 17, 9, 25, 35, 33, 30, 25, 20, 14, 7,
      17, 24, 23, 20, 17, 14, 9, 5, 9, 12, 12, 10, 9, 7, 5, 2]

track = {"max_operand": 0}


def rs(x):
    return (x + 128) >> 8


def mul(x, c):
    track["max_operand"] = max(track["max_operand"], abs(x))
    return x * c


def one_d(s, final):
    b1 = s[4]; b3 = s[2] + s[6]; b4 = s[5] - s[3]; tmp1 = s[1] + s[7]; tmp2 = s[3] + s[5]
    b6 = s[1] - s[7]; b7 = tmp1 + tmp2; m0 = s[0]
    x4 = rs(mul(b6, 473) - mul(b4, 196)) - b7
    x0 = x4 - rs(mul(tmp1 - tmp2, 362))
    x1 = m0 - b1
    x2 = rs(mul(s[2] - s[6], 362)) - b3
    x3 = m0 + b1
    y3 = x1 + x2; y4 = x3 + b3; y5 = x1 - x2; y6 = x3 - b3
    y7 = -x0 - rs(mul(b4, 473) + mul(b6, 196))
    out = [b7 + y4, x4 + y3, y5 - x0, y6 - y7, y6 + y7, x0 + y5, y3 - x4, y4 - b7]
    return [rs(v) if final else v for v in out]


def idct(block):
    v = list(block)
    for i in range(8):
        col = one_d([v[8 * k + i] for k in range(8)], False)
        for k in range(8):
            v[8 * k + i] = col[k]
    for i in range(0, 64, 8):
        v[i:i + 8] = one_d(v[i:i + 8], True)
    return v


def main():
    resp = np.zeros((64, 64))
    inter = np.zeros(64)
    for k in range(64):
        amp = 2048 * PM[k] + (4096 << 8 if k == 0 else 0)
        blk = [0] * 64
        blk[k] = amp
        track["max_operand"] = 0
        resp[k] = np.abs(idct(blk))
        inter[k] = track["max_operand"]
    out_bound = resp.sum(axis=0).max() + 64
    # operands of the multiplications: sum of single-input maxima bounds any combination (triangle inequality)
    op_bound = inter.sum() + 64 * 8
    print("max |output| over all inputs      <= %d (int16 needs < 32768)" % out_bound)
    print("max |multiplication operand|      <= %d (24-bit signed needs < %d)" % (op_bound, 1 << 23))
    assert out_bound < 32768 and op_bound < (1 << 23)
    return out_bound, op_bound


if __name__ == "__main__":
    main()
