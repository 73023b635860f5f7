"""sha1 over the kernel and host sources (onnxstream_amd/csrc/**): the identity of the tree a counter file was collected on.  bench.py reports counter-derived
fields only from a profiles/r04_pmc*.json whose `src_sha1` equals the hash of the tree it runs from (a GPU box has no .git: a commit id is not available there)."""
import hashlib
import os


def src_sha1(repo: str) -> str:
    h = hashlib.sha1()
    root = os.path.join(repo, "onnxstream_amd", "csrc")
    for dp, dn, fn in sorted(os.walk(root)):
        dn.sort()
        if os.path.basename(dp) == "build":
            continue
        for f in sorted(fn):
            if f.endswith((".hip", ".h", ".cpp")):
                h.update(f.encode())
                h.update(open(os.path.join(dp, f), "rb").read())
    return h.hexdigest()


if __name__ == "__main__":
    print(src_sha1(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
