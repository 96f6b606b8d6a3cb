#!/usr/bin/env python3
"""Seeded synthetic FASTA generator.

i.i.d. uniform over {A,C,G,U}, fixed (or ranged) length, 60-column FASTA, names
``<prefix><i>``.  Python ``random.Random(seed).choice('ACGU')`` per base, which is the
generator the BASELINE.md measurements were made with: DB seed 1, query seed 2.

`gen` draws base by base through Python's `random` (the definition).  `gen_fixed` produces the
same sequences for a fixed length two orders of magnitude faster: it replays what
`Random(seed).choice('ACGU')` does on the Mersenne Twister stream with numpy (CPython's
`choice` -> `_randbelow(4)` draws `getrandbits(3)` = the top three bits of one 32-bit output and
rejects values >= 4; the generator state is taken from `Random(seed).getstate()`).  tests/test_host.py checks that the two agree; on any mismatch of the first
sequence (a different CPython or numpy) `gen_fixed` falls back to `gen`.
"""
import argparse
import random
import sys


def gen(n, length, seed, prefix, alphabet="ACGU", max_length=None):
    rng = random.Random(seed)
    for i in range(n):
        L = length if max_length is None else rng.randint(length, max_length)
        yield f"{prefix}{i}", "".join(rng.choice(alphabet) for _ in range(L))


def _fixed_fast(n, length, seed, alphabet):
    import numpy as np
    st = random.Random(seed).getstate()[1]  # CPython's own seeding: 624 words + position
    bg = np.random.MT19937()
    bg.state = {"bit_generator": "MT19937", "state": {"key": np.array(st[:624], dtype=np.uint32), "pos": int(st[624])}}
    rs = np.random.RandomState(bg)  # .bytes() = the raw 32-bit outputs, little-endian
    lut = np.frombuffer(alphabet.encode(), dtype=np.uint8)
    need = n * length
    out = np.empty(need, np.uint8)
    have = 0
    while have < need:
        draws = min(max(2 * (need - have) + 1024, 4096), 1 << 26)
        r = np.frombuffer(rs.bytes(4 * draws), dtype=np.uint8)[3::4] >> 5  # getrandbits(3) of each output
        r = r[r < 4][:need - have]  # (nothing is drawn after the last base, so the stream position is free)
        out[have:have + len(r)] = lut[r]
        have += len(r)
    return out.reshape(n, length)


def gen_fixed(n, length, seed, prefix, alphabet="ACGU"):
    """Same records as gen(n, length, seed, prefix) (fixed length), as a list."""
    if n <= 0 or length <= 0 or len(alphabet) != 4:
        return list(gen(n, length, seed, prefix, alphabet))
    try:
        arr = _fixed_fast(n, length, seed, alphabet)
        first = next(gen(1, length, seed, prefix, alphabet))[1]
        if arr[0].tobytes().decode() != first:
            raise ValueError("stream mismatch")
    except Exception:  # a numpy without the legacy seeding hook, or a different CPython algorithm
        return list(gen(n, length, seed, prefix, alphabet))
    return [(f"{prefix}{i}", arr[i].tobytes().decode()) for i in range(n)]


def write_fasta(path, records, width=60):
    with open(path, "w") as f:
        for name, seq in records:
            f.write(f">{name}\n")
            for k in range(0, len(seq), width):
                f.write(seq[k:k + width] + "\n")


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__)
    ap.add_argument("-n", type=int, required=True, help="number of sequences")
    ap.add_argument("-L", type=int, required=True, help="sequence length (minimum if --max-length)")
    ap.add_argument("--max-length", type=int, default=None)
    ap.add_argument("--seed", type=int, required=True)
    ap.add_argument("--prefix", default="q")
    ap.add_argument("-o", required=True)
    a = ap.parse_args(argv)
    if a.max_length is None:
        write_fasta(a.o, gen_fixed(a.n, a.L, a.seed, a.prefix))
    else:
        write_fasta(a.o, gen(a.n, a.L, a.seed, a.prefix, max_length=a.max_length))


if __name__ == "__main__":
    sys.exit(main())
