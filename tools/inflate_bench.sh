#!/bin/bash
# host/fast_inflate.cpp on this machine: a FASTQ-like text of $1 MB (default 400) deflated at level 6, decoded by one thread and by 2 .. 16
# (tests/cpp/inflate_harness.cpp par), and by zlib (time).  No GPU involved; run on the GPU box for its host's numbers.
set -u
MB=${1:-400}
T=${TMPDIR:-/tmp}
g++ -O3 -std=c++17 -pthread -o $T/inflate_harness tests/cpp/inflate_harness.cpp biokanga_amd/csrc/host/fast_inflate.cpp -lz || exit 1
python3 - "$MB" "$T/bench.deflate" <<'PY'
import sys, zlib, numpy as np
mb, path = int(sys.argv[1]), sys.argv[2]
rng = np.random.default_rng(5)
n = mb * (1 << 20) // 250
L = 100
rec = np.empty((n, 29 + 2 * L), dtype=np.uint8)
ids = np.char.zfill(np.arange(n).astype(str), 9)
rec[:, :15] = np.frombuffer(b"@SRR0000001.000", dtype=np.uint8)[:15]
rec[:, 15:24] = np.frombuffer("".join(ids).encode(), dtype=np.uint8).reshape(n, 9)
rec[:, 24] = 10
rec[:, 25:25 + L] = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=(n, L))]
rec[:, 25 + L:28 + L] = np.frombuffer(b"\n+\n", dtype=np.uint8)
q = np.minimum(73, 33 + np.abs(rng.normal(36, 3, size=(n, L)))).astype(np.uint8)
run = rng.random((n, L)) < 0.6                       # real scores come in runs
for k in range(1, L):
    q[:, k] = np.where(run[:, k], q[:, k - 1], q[:, k])
rec[:, 28 + L:28 + 2 * L] = q
rec[:, 28 + 2 * L] = 10
z = zlib.compressobj(6, zlib.DEFLATED, -15)
data = rec.tobytes()
with open(path, "wb") as f:
    for o in range(0, len(data), 64 << 20):
        f.write(z.compress(data[o:o + (64 << 20)]))
    f.write(z.flush())
print("text", len(data), "bytes")
PY
ls -l $T/bench.deflate
$T/inflate_harness time $T/bench.deflate $((MB * 1100000))
for n in 2 4 8 12 16; do $T/inflate_harness par $T/bench.deflate $n; done
rm -f $T/bench.deflate
