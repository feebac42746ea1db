#!/usr/bin/env python3
"""Generates tests/golden/*.json|npz from the CPU oracle.

Provenance: the reference (Rust) cannot be built or run in this image, and it has no tests or golden
outputs of its own, so these fixtures are produced by oracle/mdbg_oracle.cpp (a line-by-line restatement,
pinned on the nthash crate's known-answer vectors).  The config-1 counts and the node-set SHA-256 were
additionally reproduced by an independent numpy restatement during the survey.
Run:  python tests/golden/make_golden.py
"""
import gzip, hashlib, json, os, sys
import numpy as np
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
from oracle import oracle as O


def read_fasta_gz(path):
    reads = []
    with gzip.open(path, "rb") as f:
        for line in f:
            if not line.startswith(b">"):
                reads.append(line.strip())
    return reads


def node_sha(keys, abundance):
    ks = sorted(tuple(int(x) for x in keys[i]) + (int(abundance[i]),) for i in range(len(abundance)))
    s = "".join(",".join(map(str, t[:-1])) + ":" + str(t[-1]) + "\n" for t in ks)
    return hashlib.sha256(s.encode()).hexdigest()


def edge_sha(r):
    es = sorted((int(a), chr(b), int(c), chr(d), int(e)) for a, b, c, d, e in
                zip(r["edge_n1"], r["edge_o1"], r["edge_n2"], r["edge_o2"], r["edge_overlap"]))
    s = "".join("L\t%d\t%s\t%d\t%s\t%dM\n" % e for e in es)
    return hashlib.sha256(s.encode()).hexdigest()


def main():
    reads = read_fasta_gz(os.path.join(HERE, "reads-0.00.fa.gz"))
    bases, offs = O.concat_reads(reads)
    k, l, d, A = 7, 10, 0.0008, 2
    sk = O.sketch(bases, offs, l, d)
    g = O.Graph(k, l, d, A)
    g.ingest(bases, offs)
    r = g.finalize()
    o = sk["off"]
    out = dict(
        config=dict(k=k, l=l, density=d, minabund=A, file="reads-0.00.fa.gz"),
        n_reads=len(reads), n_bases=int(offs[-1]), hash_bound=O.hash_bound(d),
        n_minimizers=int(len(sk["hashes"])), n_windows=r["n_windows"], n_nodes_before=r["n_nodes_before"],
        n_nodes=r["n_nodes"], n_edges=r["n_edges"], presimp_removed=r["presimp_removed"],
        read0_n=int(o[1] - o[0]), read0_first3=[[int(sk["pos"][i]), int(sk["hashes"][i])] for i in range(3)],
        read1_n=int(o[2] - o[1]), read1_first=[int(sk["pos"][o[1]]), int(sk["hashes"][o[1]])],
        minimizers_sha256=hashlib.sha256(sk["hashes"].tobytes() + sk["pos"].tobytes() + sk["off"].tobytes()).hexdigest(),
        nodes_sha256=node_sha(r["keys"], r["abundance"]),
        nodes_full_sha256=hashlib.sha256(b"".join(r[f].tobytes() for f in
                                         ("keys", "index", "abundance", "seqlen", "shift", "shift_full", "src_read", "src_start", "src_end", "reversed"))).hexdigest(),
        edges_sha256=edge_sha(r), abundance_min=int(r["abundance"].min()), abundance_max=int(r["abundance"].max()),
    )
    with open(os.path.join(HERE, "example_cfg1.json"), "w") as f:
        json.dump(out, f, indent=1)
    # full node + edge table of config 1 (small: 104 nodes) for field-by-field comparisons
    np.savez_compressed(os.path.join(HERE, "example_cfg1_nodes.npz"),
                        **{f: r[f] for f in ("keys", "index", "abundance", "seqlen", "shift", "shift_full", "src_read", "src_start",
                                             "src_end", "reversed", "edge_n1", "edge_n2", "edge_o1", "edge_o2", "edge_overlap")},
                        hashes=sk["hashes"], pos=sk["pos"], off=sk["off"])
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
