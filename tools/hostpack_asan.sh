#!/bin/bash
# AddressSanitizer + UBSan over the C host loops of riser_amd/csrc/hostpack.c (CPU only; GPU ASan is not available on this
# pool): builds an instrumented _hostpack into a scratch directory and drives every entry point, error paths included.
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$(mktemp -d)
mkdir -p "$OUT/riser_amd"
gcc -O1 -g -shared -fPIC -Wall -pthread -fsanitize=address,undefined -fno-omit-frame-pointer \
    -I"$(python3 -c 'import sysconfig; print(sysconfig.get_paths()["include"])')" \
    "$ROOT/riser_amd/csrc/hostpack.c" -o "$OUT/_hostpack.so"
cat > "$OUT/drive.py" <<'PY'
import importlib.util, sys, numpy as np
spec = importlib.util.spec_from_file_location("_hostpack", sys.argv[1]); hp = importlib.util.module_from_spec(spec); spec.loader.exec_module(hp)
class R:
    def __init__(self, i, a): self.id, self.raw_data = i, (a if isinstance(a, memoryview) else a.tobytes())
rng = np.random.default_rng(3)
reads = [R(f"id-{i}", rng.integers(-300, 4000, size=int(rng.integers(0, 5000)), dtype=np.int16)) for i in range(300)]
reads.append(R("mv", memoryview(np.arange(90, dtype=np.int16).tobytes())[20:120]))
lens = np.empty(len(reads), dtype=np.int64); hp.lengths(reads, lens)
for start in (np.zeros(len(reads), dtype=np.int64), np.minimum(lens, rng.integers(0, 60, size=len(reads)))):
    out = np.empty(int((lens - start).sum()), dtype=np.int16); assert hp.gather(reads, start, out) == out.size
big = [R(f"b{i}", rng.integers(0, 100, size=40000, dtype=np.int16)) for i in range(200)]
out = np.empty(200 * 40000, dtype=np.int16); hp.gather(big, np.zeros(200, dtype=np.int64), out)          # the threaded path
import threading
def many():                                                       # the persistent copy pool, re-used and contended
    o2 = np.empty(200 * 40000, dtype=np.int16)
    for _ in range(6): assert hp.gather(big, np.zeros(200, dtype=np.int64), o2) == o2.size
    assert np.array_equal(o2[:40000], np.frombuffer(big[0].raw_data, dtype=np.int16))
ts = [threading.Thread(target=many) for _ in range(3)]; [t.start() for t in ts]; many(); [t.join() for t in ts]
dl = hp.decided(reads, np.arange(len(reads), dtype=np.int64), np.arange(len(reads), dtype=np.int64), (np.arange(len(reads)) % 4).astype(np.uint8), (2, 1, 3))
assert sum(map(len, dl)) == sum(1 for i in range(len(reads)) if i % 4) and dl[0][0] == (2, "id-2")
# store_slice: new reads, then the same reads 100 samples longer (delta path), then other samples under the same ids (mismatch)
nr = 64; T = 32
rid = np.full(nr, None, dtype=object); rhave = np.zeros(nr, dtype=np.int64); rtail = np.zeros((nr, T), dtype=np.int16)
sigs = [rng.integers(-300, 4000, size=int(rng.integers(40, 3000)), dtype=np.int16) for _ in range(nr)]
for turn in range(3):
    if turn == 1: sigs = [np.concatenate([s_, rng.integers(0, 9, size=100, dtype=np.int16)]) for s_ in sigs]
    if turn == 2: sigs = [rng.integers(0, 9, size=s_.size, dtype=np.int16) for s_ in sigs]
    rr = [R(f"s{i}", s_) for i, s_ in enumerate(sigs)]
    idsv = np.empty(nr, dtype=object); idsv[:] = [r_.id for r_ in rr]
    ln = np.array([s_.size for s_ in sigs], dtype=np.int64)
    st_, cd_, stt = np.empty(nr, dtype=np.int64), np.empty(nr, dtype=np.uint8), np.zeros(3, dtype=np.int64)
    stg = np.empty(int(ln.sum()), dtype=np.int16)
    tot = hp.store_slice(rr, idsv, rid, np.arange(nr, dtype=np.int64), ln, (ln <= 2900).astype(np.uint8), rhave, rtail, T, stg, st_, cd_, stt)
    assert tot == int((ln - st_).sum())
    assert (turn != 1) or stt[1] > 50, stt
    assert (turn != 2) or (stt[1] == 0 and stt[2] > 50), stt
assert hp.repr_double(0.1) == repr(0.1) and hp.repr_double(5e-324) == "5e-324" and hp.repr_double(float("nan")) == "nan"
ids = np.empty(len(reads), dtype=object); ids[:] = hp.attrs(reads, "id")
o = np.zeros(len(reads), dtype=np.int64)
hp.lookup({r.id: i for i, r in enumerate(reads[::2])}, ids, o); hp.lookup({}, list(ids), o); hp.lookup({"a": 1}, ids[3:9], o)
ch = np.empty(len(reads), dtype=np.int64); hp.unpack([(i, r) for i, r in enumerate(reads)], ch)
for bad in (lambda: hp.unpack([(1,)], ch), lambda: hp.gather(reads, np.full(len(reads), 10 ** 6, dtype=np.int64), out),
            lambda: hp.lookup({"id-0": "x"}, ids[:1], o), lambda: hp.attrs(reads, "nope")):
    try: bad()
    except (TypeError, ValueError, AttributeError): pass
s = hp.format_rows("1,", reads, np.arange(50, dtype=np.int64), np.arange(50, dtype=np.int64), np.arange(50, dtype=np.int32), ",a;b;c,",
                   rng.random((50, 3)), 3, ",0.9,enrich,", rng.integers(0, 4, size=50).astype(np.uint8), ("try_again", "accept", "reject", "no_decision"))
assert s.count("\n") == 50
print("hostpack under ASan + UBSan: ok")
PY
LD_PRELOAD=$(gcc -print-file-name=libasan.so) ASAN_OPTIONS=detect_leaks=0 python3 "$OUT/drive.py" "$OUT/_hostpack.so"
rm -rf "$OUT"
