"""sha256 over the kernel sources (csrc/*.hip, csrc/*.h, include/*.h): ties committed PMC numbers to the code they measured."""
import glob, hashlib, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def kernel_source_hash() -> str:
    h = hashlib.sha256()
    files = sorted(glob.glob(os.path.join(ROOT, "flash_hash_join_amd", "csrc", "*.hip")) + glob.glob(os.path.join(ROOT, "flash_hash_join_amd", "csrc", "*.h"))
                   + glob.glob(os.path.join(ROOT, "include", "*.h")))
    for f in files:
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()


if __name__ == "__main__":
    print(kernel_source_hash())
