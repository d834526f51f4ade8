"""Re-key profiles/hbm_traffic.json on the current library sources from a set of scripts/gpu_round.sh.
    python3 scripts/update_traffic.py r05_v1      (reads gpurun_out/r05_v1/pmc_summary.txt; the set must have been copied to
                                                   profiles/r05_v1_pmc_summary.txt, which the entry names as its source)"""
import ast, json, os, re, subprocess, sys
tag = sys.argv[1]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
vals = {}
for line in open(os.path.join(root, "gpurun_out", tag, "pmc_summary.txt")):
    if not line.startswith("mf::kf_chunk_lds_kernel<double, 6, 1, true, false>"):
        continue
    vals.update(ast.literal_eval(line[line.index("{"):]))
fetch, write = vals["FETCH_SIZE"], vals["WRITE_SIZE"]
path = os.path.join(root, "profiles", "hbm_traffic.json")
doc = json.load(open(path))
key = "kf_loglik B=1024 T=10000 d=6 m=1 f64"
prev = doc[key]
sha = subprocess.check_output([sys.executable, os.path.join(root, "scripts", "csrc_hash.py")], text=True).strip()
ver = int(re.search(r"mf_version\(void\)\s*\{\s*return\s+(\d+)", open(os.path.join(root, "markovflow_amd", "csrc", "mf_api.hip")).read()).group(1))
doc[key] = {"kernel": prev["kernel"], "round": f"{tag} (scripts/gpu_round.sh {tag}, the library this hash names)", "library_version": ver,
            "csrc_sha256": sha, "FETCH_SIZE_KiB": round(fetch, 1), "WRITE_SIZE_KiB": round(write, 1),
            "traffic_bytes": int(round(2 * fetch * 1024 + write * 1024)), "source": f"profiles/{tag}_pmc_summary.txt", "previous": prev}
json.dump(doc, open(path, "w"), indent=1)
print(key, doc[key]["traffic_bytes"], "bytes,", round(doc[key]["traffic_bytes"] / 6963200000, 3), "x algorithmic; csrc", sha)
