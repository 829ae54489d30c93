#!/bin/bash
# usage (on the GPU box, from the repo root): tools/gpu_pmc_final.sh <outfile> <op> [<op> ...] — per op of tools/run_op.py: one --kernel-trace pass (20 calls)
# for the time and three --pmc passes (FETCH_SIZE; WRITE_SIZE; SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES), each with --kernel-trace only; one line per kernel.
out=$1; shift
export TMPDIR=/tmp
for op in "$@"; do
  d=/tmp/pf_$$; rm -rf $d; mkdir -p $d
  timeout 300 rocprofv3 --kernel-trace -d $d/kt -o r -- python tools/run_op.py $op 20 > $d/kt.log 2>&1
  for c in FETCH_SIZE WRITE_SIZE "SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES"; do
    n=$(echo $c | cut -d' ' -f1)
    timeout 300 rocprofv3 --kernel-trace --pmc $c -d $d/$n -o r -- python tools/run_op.py $op 5 > $d/$n.log 2>&1
  done
  python - $d $op >> $out <<'PY'
import sqlite3, sys, glob, collections, re
d, op = sys.argv[1], sys.argv[2]
def db(n):
    g = glob.glob(f"{d}/{n}/**/*.db", recursive=True)
    return sqlite3.connect(g[0]) if g else None
print(f"== tools/run_op.py {op}")
kt = db("kt")
times = {}
if kt:
    for n, cnt, avg in kt.execute("select s.kernel_name, count(*), avg(d.end - d.start) from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id = s.id group by s.kernel_name"):
        times[n] = (cnt, avg / 1e3)
    def key_mangled(n):  # _ZN2zg<len><ident>I<literal args>E... -> "ident<a, b, ...>"
        m = re.match(r"_ZN2zg(\d+)", n)
        if not m: return n
        p0 = m.end(); ident = n[p0:p0 + int(m.group(1))]; rest = n[p0 + int(m.group(1)):]
        args = []
        if rest.startswith("I"):
            for kind, neg, val in re.findall(r"L([ibjm])(n?)(\d+)E", rest[1:rest.find("EE") + 1] if "EE" in rest else rest[1:]):
                args.append(("true" if val == "1" else "false") if kind == "b" else ("-" if neg else "") + val)
        return ident + ("<" + ", ".join(args) + ">" if args else "")
    times = {key_mangled(n): v for n, v in times.items()}
vals = collections.defaultdict(dict)
for n in ("FETCH_SIZE", "WRITE_SIZE", "SQ_INSTS_VALU"):
    c = db(n)
    if not c: continue
    acc = collections.defaultdict(list)
    for k, name, v in c.execute("select kernel_name, counter_name, value from counters_collection"): acc[(k, name)].append(v)
    for (k, name), vs in acc.items(): vals[k][name] = sum(vs) / len(vs)
for k in sorted(vals, key=lambda k: -times.get(k, (0, 0))[1]):
    if "zg" not in k: continue
    kk = re.sub(r"^(void )?zg::", "", re.sub(r"[(].*", "", k))
    v = vals[k]; cnt, us = times.get(kk, (0, float("nan")))
    rd = 2 * v.get("FETCH_SIZE", 0) * 1024 / 1e6  # gfx950: FETCH_SIZE counts 32-byte requests in 64-byte units (MI355X_MICROARCH.md): x 2; KiB units
    wr = v.get("WRITE_SIZE", 0) * 1024 / 1e6
    print(f"  {re.sub(r'[(].*', '', k)[:66]:66s} {us:8.1f} us x{cnt / 20:4.1f}/call  read {rd:8.1f} MB  written {wr:8.1f} MB  = {(rd + wr) / us:5.2f} TB/s  VALU {v.get('SQ_INSTS_VALU', 0) / 1e6:7.2f} M  SALU {v.get('SQ_INSTS_SALU', 0) / 1e6:6.2f} M  waves {int(v.get('SQ_WAVES', 0)):7d}")
PY
  rm -rf $d
done
