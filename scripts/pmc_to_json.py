"""Turn the two rocprofv3 --pmc passes of scripts/ffn_pmc_probe.py (FETCH_SIZE and WRITE_SIZE, collected separately)
into profiles/ffn_block_pmc.json -- the file bench.py reads ``roofline.traffic`` from.

Usage: python scripts/pmc_to_json.py <fetch counter_collection.csv> <write counter_collection.csv> <source tag> [out.json]

Per /opt/skills/guides/MI355X_MICROARCH.md (HBM / rocprofv3 section): both counters are in KB per dispatch, and on gfx950
FETCH_SIZE under-reports wide coalesced streaming reads by 2x, so traffic = 2 * FETCH_SIZE + WRITE_SIZE.
The four launches a decoder feed-forward block issues per step are picked out by kernel name; the probe issues each 30
times and the per-dispatch mean is recorded.
"""
import collections
import csv
import json
import os
import sys

# (label, substring(s) that identify the dispatch, algorithmic bytes) -- M = 6528 tokens, 128 <-> 1024, bf16 storage
M, C, F = 32 * 204, 128, 1024
WB = 2 * 2 * C * F
LAUNCHES = [
    ("ffn_pair forward", ("ffn_pair_kernel<false",), 2 * M * C + WB + 2 * M * F + 8 * M * C),
    ("ffn_pair input gradients", ("ffn_pair_kernel<true",), 4 * M * C + 2 * M * F + WB + 2 * M * F + 2 * M * C),
    # bgemm_tn_kernel<A_F32, B_F32, BN, BK>: the probe's two weight gradients are <true, false> (fp32 dy) and <false, false>
    ("wgrad 128x1024 (fp32 dy x bf16 hidden)", ("bgemm_tn_kernel<true,false,",), 4 * M * C + 2 * M * F + 4 * C * F),
    ("wgrad 1024x128 (bf16 dz x bf16 ln-out)", ("bgemm_tn_kernel<false,false,",), 2 * M * F + 2 * M * C + 4 * C * F),
]


def means(path, counter):
    tot = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        a = tot[r["Kernel_Name"]]
        a[0] += 1
        a[1] += float(r["Counter_Value"])
    return {k: (v[1] / v[0], v[0]) for k, v in tot.items()}


def pick(table, keys):
    hits = [(n, v) for n, v in table.items() if any(k in n.replace(" ", "") for k in keys)]
    if len(hits) != 1:
        raise SystemExit("expected exactly one kernel matching %r, found %r (kernels: %r)" %
                         (keys, [h[0][:60] for h in hits], sorted(k[:60] for k in table)))
    return hits[0][1]


def main():
    fetch, write, tag = means(sys.argv[1], "FETCH_SIZE"), means(sys.argv[2], "WRITE_SIZE"), sys.argv[3]
    out = sys.argv[4] if len(sys.argv) > 4 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                                              "profiles", "ffn_block_pmc.json")
    launches, tot, tot_alg = {}, 0.0, 0.0
    for label, keys, alg in LAUNCHES:
        (f_kb, n_f), (w_kb, n_w) = pick(fetch, keys), pick(write, keys)
        traffic = (2.0 * f_kb + w_kb) * 1024.0
        launches[label] = {"fetch_size_kb": round(f_kb, 1), "write_size_kb": round(w_kb, 1), "dispatches": [n_f, n_w],
                           "traffic_bytes": round(traffic), "algorithmic_bytes": alg}
        tot += traffic
        tot_alg += alg
    doc = {}
    if os.path.exists(out):
        doc = json.load(open(out))
    doc["bf16"] = {"source": tag, "formula": "2 * FETCH_SIZE + WRITE_SIZE (KB per dispatch, gfx950 correction)",
                   "mean_traffic_bytes": round(tot / len(LAUNCHES)), "mean_algorithmic_bytes": round(tot_alg / len(LAUNCHES)),
                   "launches": launches}
    json.dump(doc, open(out, "w"), indent=1)
    print(json.dumps(doc["bf16"], indent=1))


if __name__ == "__main__":
    main()
