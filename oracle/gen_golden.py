#!/usr/bin/env python3
"""oracle/gen_golden.py -- regenerate tests/golden/ from the REFERENCE (oracle/_ref).

TEST INFRASTRUCTURE.  Runs only where /root/reference exists (oracle/_ref built by
`make -C oracle ref`).  Everything written is an output of the reference's own
classes: the seeded generator (jellyfish/generate_sequence.cc) for the inputs and
count/dump/histo/stats (ref_jf) for the expected results.  The fixtures are small so
they travel with the repo; the GPU box never needs /root/reference.
"""
import json
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF_JF = os.path.join(HERE, "_ref", "ref_jf")
REF_GEN = os.path.join(HERE, "_ref", "ref_generate_sequence")
OUT = os.path.join(ROOT, "tests", "golden")


def run(cmd, **kw):
    return subprocess.run(cmd, check=True, capture_output=True, **kw).stdout.decode()


def main():
    if not (os.access(REF_JF, os.X_OK) and os.access(REF_GEN, os.X_OK)):
        sys.exit("oracle/_ref missing: run `make -C oracle ref` where /root/reference exists")
    os.makedirs(OUT, exist_ok=True)
    env = dict(os.environ, SOURCE_DATE_EPOCH="0")   # reproducible header provenance
    manifest = {"generator": "oracle/gen_golden.py", "cases": []}

    # inputs: the reference's generator, same flavour as BASELINE's reads150 (150-base records, 70 columns)
    run([REF_GEN, "-s", "42", "-r", "150", "-o", os.path.join(OUT, "reads150_s42"), "6000"])
    run([REF_GEN, "-s", "1473540700", "-q", "-o", os.path.join(OUT, "reads_fq_s1473540700"), "3000"])
    fa = os.path.join(OUT, "reads150_s42.fa")
    fq = os.path.join(OUT, "reads_fq_s1473540700.fq")
    # hand-made edge cases: lower case, N / IUPAC resets, CRLF, blank lines, empty record, short record
    edge = os.path.join(OUT, "edge_cases.fa")
    with open(edge, "wb") as f:
        f.write(b">r1 mixed case and N\r\nACGTacgtACGTacgtACGTNNacgtACGTACGTACGTACGTAC\r\nGGGGTTTTAAAACCCC\r\n"
                b">r2 empty\n>r3 short\nACGT\n>r4 iupac\nACGTRYACGTACGTACGTACGTACGTACGTA-CGTACGTACGTACGTACGTAC\n\n\n"
                b"TTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTT\n>r5 palindromes\nACGTACGTACGTACGTACGTACGTACGTACGTACGTACGT\n")

    cases = [
        ("reads150_k21C", fa, 21, True, "64k"),
        ("reads150_k21", fa, 21, False, "64k"),
        ("reads150_k15C", fa, 15, True, "64k"),
        ("reads150_k31C", fa, 31, True, "64k"),
        ("reads150_k32", fa, 32, False, "64k"),
        ("reads150_k5C", fa, 5, True, "64k"),
        ("reads150_k33C", fa, 33, True, "64k"),
        ("reads150_k63C", fa, 63, True, "64k"),
        ("reads150_k64", fa, 64, False, "64k"),
        ("edge_k40C", edge, 40, True, "4k"),
        ("fastq_k21C", fq, 21, True, "64k"),
        ("edge_k8C", edge, 8, True, "4k"),
        ("edge_k21C", edge, 21, True, "4k"),
    ]
    for name, path, k, canonical, size in cases:
        jf = os.path.join(OUT, name + ".ref.jf")
        cmd = [REF_JF, "count", "-m", str(k), "-s", size, "-t", "2", "-o", jf, path]
        if canonical:
            cmd.insert(2, "-C")
        subprocess.run(cmd, check=True, env=env, cwd=OUT)
        dump = sorted(run([REF_JF, "dump", "-c", jf]).splitlines())
        with open(os.path.join(OUT, name + ".dump"), "w") as f:
            f.write("\n".join(dump) + ("\n" if dump else ""))
        with open(os.path.join(OUT, name + ".histo"), "w") as f:
            f.write(run([REF_JF, "histo", jf]))
        with open(os.path.join(OUT, name + ".stats"), "w") as f:
            f.write(run([REF_JF, "stats", jf]))
        keep_jf = name in ("reads150_k21C", "edge_k8C")   # two reference-written files for the read-side tests
        if not keep_jf:
            os.unlink(jf)
        manifest["cases"].append({"name": name, "input": os.path.basename(path), "k": k, "canonical": canonical,
                                  "size": size, "ref_jf": os.path.basename(jf) if keep_jf else None,
                                  "distinct": len(dump)})
    # ---- config 3: Bloom counter first pass (jellyfish bc) and count --bc ------------------------
    # input in which roughly half of the reads occur twice, so the filter really separates k-mers
    lines = open(fa).read().splitlines()
    recs = [lines[i:i + 4] for i in range(0, len(lines), 4)]       # header + 3 sequence lines per 150-base read
    dup = os.path.join(OUT, "reads150_dup.fa")
    with open(dup, "w") as f:
        for r in recs + recs[: len(recs) // 2]:
            f.write("\n".join(r) + "\n")
    for name, k, canonical in (("bc_k21C", 21, True), ("bc_k31", 31, False)):
        bcf = os.path.join(OUT, name + ".ref.bc")
        cmd = [REF_JF, "bc", "-m", str(k), "-s", "9000", "-f", "0.001", "-t", "2", "-o", bcf, dup]
        if canonical:
            cmd.insert(2, "-C")
        subprocess.run(cmd, check=True, env=env, cwd=OUT)
        jf = os.path.join(OUT, name + ".filtered.jf")
        cmd = [REF_JF, "count", "-m", str(k), "-s", "64k", "-t", "2", "--bc", bcf, "-o", jf, dup]
        if canonical:
            cmd.insert(2, "-C")
        subprocess.run(cmd, check=True, env=env, cwd=OUT)
        d = sorted(run([REF_JF, "dump", "-c", jf]).splitlines())
        with open(os.path.join(OUT, name + ".filtered.dump"), "w") as f:
            f.write("\n".join(d) + "\n")
        os.unlink(jf)
        manifest.setdefault("bloom", []).append({"name": name, "input": "reads150_dup.fa", "k": k, "canonical": canonical,
                                                 "n": 9000, "fpr": 0.001, "ref_bc": name + ".ref.bc", "kept": len(d)})

    # the reference's own golden md5s for this path (tests/parallel_hashing.sh:7-19), reproduced by
    # oracle/_ref at survey/build time; tests/test_oracle.py re-checks them whenever oracle/_ref exists
    manifest["reference_md5"] = {
        "source": "tests/parallel_hashing.sh:7-19 (inputs: tests/generate_sequence.sh:6-7)",
        "seq10m": ["-s", "3141592653", "10000000"],
        "seq1m": ["-s", "1040104553", "1000000", "1000000", "1000000", "1000000", "1000000"],
        "m15_s2M.histo": "864c0b0826854bdc72a85d170549b64b",
        "m15.stats": "41fd8408dde0ea14bec7425b1a877140",
        "binary.dump": "376761a6e273b57b3428c14e3b536edf",
        "binary.histo": "9251799dd5dbd3f617124aa2ff72112a",
        "binary.stats": "c30cba4fe2886cea4abb27f5c30ea35e",
        "m15_s2M_L2_U3.histo": "94625cd2d59e278f08421a673eb0926a",
        "query_one_count": "45fb383344e0fb0b7540718339be4c03",
        "bloom_counter_noop.histo": "9251799dd5dbd3f617124aa2ff72112a",
    }
    with open(os.path.join(OUT, "manifest.json"), "w") as f:
        json.dump(manifest, f, indent=1)
    print("wrote", OUT)


if __name__ == "__main__":
    main()
