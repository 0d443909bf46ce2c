"""Dev tool (CPU only): brute-force search of the LDS image layout of ro_conv2_split_kernel (csrc/ro_conv2.hip).

A B fragment of v_mfma_f32_16x16x32_f16 is one ds_read_b128 per lane: lane (pixel li = lane & 15, k group kq = lane >> 4) reads 16-B
chunk kq of its pixel's row.  A wave64 ds_read_b128 is served in four groups of 16 lanes (MI355X_MICROARCH.md, LDS), one LDS cycle per
group when the 16 lanes hit 16 distinct 16-B slots of the 256-B bank row.  The tiles of these convs are 16 consecutive OUTPUT pixels of
a W_out-wide raster read from a W_in-wide input image (+ tap offsets), so consecutive lanes are NOT consecutive rows at a row wrap.
Searched: image pitch P (pixels), XOR key = ((X + c Y + e) >> sh) & 3 on the chunk index, 64-B rows (32 channels per plane).
Result (cycles per read, ideal 4.0): only P = W_in + 2 with c = W_out mod 8, sh = 1 is conflict-free; plain rows cost 6.7 - 10."""
groups = [[0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27], [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31]]
groups += [[l + 32 for l in g] for g in groups]


def cycles(addrs):
    tot = 0
    for g in groups:
        slots = {}
        for l in g:
            a = addrs[l]
            slots.setdefault((a // 16) % 16, set()).add(a // 16)
        tot += max(len(v) for v in slots.values())
    return tot


def evaluate(W_in, H_in, W_out, H_out, ncrop, P, c, sh, e=0, gk=0):
    npx = W_out * H_out * ncrop
    tot = n = worst = 0
    for t0 in range(0, npx, 16):
        for kh in range(3):
            for kw in range(3):
                addrs = []
                for l in range(64):
                    li, kq = l & 15, l >> 4
                    q = min(t0 + li, npx - 1)
                    g, qq = divmod(q, W_out * H_out)
                    y, x = divmod(qq, W_out)
                    Y, X = y + kh, x + kw
                    r = (g * H_in + Y) * P + X
                    key = ((X + c * Y + e + gk * g) >> sh) & 3
                    addrs.append(r * 64 + ((kq ^ key) * 16))
                cy = cycles(addrs); tot += cy; n += 1; worst = max(worst, cy)
    return round(tot / n, 3), worst


if __name__ == "__main__":
    cases = {"R-Net conv2, one crop (11x11 -> 9x9)": (11, 11, 9, 9, 1), "R-Net conv2, three crops": (11, 11, 9, 9, 3),
             "O-Net conv2 band (13 x 23 -> 11 x 21)": (23, 13, 21, 11, 1)}
    for name, (wi, hi, wo, ho, nc) in cases.items():
        print(name)
        for P in range(wi, wi + 5):
            row = []
            for sh in (0, 1, 2):
                row.append(evaluate(wi, hi, wo, ho, nc, P, wo % (4 << sh), sh, 0, (wo * ho) % 8 if nc > 1 else 0))
            print(f"   pitch {P}: key shift 0 / 1 / 2 -> (avg cycles, worst) {row}")
