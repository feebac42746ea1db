#!/usr/bin/env python3
"""Vectors produced by the REFERENCE'S OWN Python utilities, run in the build container (they cannot travel; the vectors can).

rust-mdbg's Rust path cannot be built here (no cargo), but two of the Python helpers it ships restate pieces of this path and run as they are:

  utils/remove_homopoly.py   the homopolymer compression of src/read.rs:157-174 as a script: a character equal to its predecessor is dropped when it is one
                             of "ACTGactgNn" (the same literal as the Rust source).  Run on one-line files -> `hpc`: (input, output) pairs.
  utils/parse_gfa.py         parse(filename): how the reference's own tools read the S lines of a .gfa (id -> KC abundance).  Run on the .gfa this
                             framework's emitter writes for BASELINE configs[0] -> `gfa_abundance`.

Nothing of those files is copied: they are executed / imported from /root/reference, and only inputs and outputs are stored
(tests/golden/reference_py_vectors.json).  Run:  python tests/golden/make_reference_py_vectors.py
"""
import gzip, importlib.util, json, os, random, subprocess, sys, tempfile
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.join(HERE, "..", "..")
REF = "/root/reference/utils"
sys.path.insert(0, ROOT)


def hpc_cases():
    rnd = random.Random(20260928)
    cases = ["A", "AA", "ACGT", "AAAACCCCGGGGTTTT", "aaAAaa", "NNNNnnnnNN", "ACGTNNACGTnnACGT", "RRYYKKMM", "AARRAAYYCC--CC", "TTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTT",
             "AaAaAa", "XXAAXX", "ACGTacgtACGT", "GGGGGGGGGGGGGGGGgggggggggggggggg", "NnNnNN", "A*A**A", "U", "UUUU", "ACGUUUACG", "0011AACC"]
    alph = ["ACGT", "ACGTN", "ACGTacgtNn", "ACGTacgtNnRYKMSWBDHVUu*-.01"]
    for i in range(60):
        a = alph[i % len(alph)]
        s = []
        for _ in range(rnd.randrange(1, 40)):
            c = rnd.choice(a)
            s.append(c * (rnd.choice([1, 1, 1, 2, 3, 5, 9, 40]) if rnd.random() < 0.6 else 1))
        cases.append("".join(s))
    return cases


def run_remove_homopoly(seq):
    with tempfile.NamedTemporaryFile("w", suffix=".txt", delete=False) as f:
        f.write(seq + "\n")
        path = f.name
    try:
        out = subprocess.run([sys.executable, os.path.join(REF, "remove_homopoly.py"), path], capture_output=True, text=True, check=True).stdout
    finally:
        os.unlink(path)
    lines = out.split("\n")
    assert lines[-1] == "" and len(lines) == 2, out
    return lines[0]


def main():
    hpc = [{"input": s, "output": run_remove_homopoly(s)} for s in hpc_cases()]
    # the .gfa of BASELINE configs[0] as this framework's host emitter writes it from the ORACLE's node table (no GPU needed), read by the reference's parser
    from oracle import oracle as O
    from rust_mdbg_amd.emit import Emitter
    reads = [ln.strip() for ln in gzip.open(os.path.join(HERE, "reads-0.00.fa.gz"), "rb") if not ln.startswith(b">")]
    b, o = O.concat_reads(reads)
    g = O.Graph(7, 10, 0.0008, 2)
    g.ingest(b, o)
    r = g.finalize(with_edges=True)
    em = Emitter()
    em.edges(r, 0.01)
    with tempfile.TemporaryDirectory() as td:
        p = os.path.join(td, "cfg1.gfa")
        em.write_gfa(p, r)
        text = open(p).read()
        spec = importlib.util.spec_from_file_location("ref_parse_gfa", os.path.join(REF, "parse_gfa.py"))
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        ab = mod.parse(p)
    out = {"provenance": "outputs of /root/reference/utils/remove_homopoly.py and utils/parse_gfa.py run in the build container by tests/golden/make_reference_py_vectors.py",
           "hpc": hpc, "gfa_text_sha256": __import__("hashlib").sha256(text.encode()).hexdigest(), "gfa_s_lines": [ln for ln in text.split("\n") if ln.startswith("S")],
           "gfa_abundance": ab}
    json.dump(out, open(os.path.join(HERE, "reference_py_vectors.json"), "w"), indent=0)
    print(len(hpc), "hpc cases,", len(ab), "S lines read by the reference's parser")


if __name__ == "__main__":
    main()
