#!/usr/bin/env python3
"""Seeded synthetic FASTA generator.

i.i.d. uniform over {A,C,G,U}, fixed (or ranged) length, 60-column FASTA, names
``<prefix><i>``.  Python ``random.Random(seed).choice('ACGU')`` per base, which is the
generator the BASELINE.md measurements were made with: DB seed 1, query seed 2.
"""
import argparse
import random
import sys


def gen(n, length, seed, prefix, alphabet="ACGU", max_length=None):
    rng = random.Random(seed)
    for i in range(n):
        L = length if max_length is None else rng.randint(length, max_length)
        yield f"{prefix}{i}", "".join(rng.choice(alphabet) for _ in range(L))


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
    write_fasta(a.o, gen(a.n, a.L, a.seed, a.prefix, max_length=a.max_length))


if __name__ == "__main__":
    sys.exit(main())
