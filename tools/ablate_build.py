"""Diagnostic variant libraries (timing experiments only, results wrong): one source re-compiled with
-DRS_ABL_* flags and linked against the shipped objects.

    python tools/ablate_build.py conv_wino4.hip nomem=RS_ABL_NOLOAD,RS_ABL_NOLDSW,RS_ABL_NOSTORE ...

writes riser_amd/lib/libabl_<name>.so (a flag written @-mllvm or @<raw> is passed to hipcc as is); run a tool against it with RISER_AMD_LIB=<path>.
"""
import os, subprocess, sys
from concurrent.futures import ThreadPoolExecutor
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from riser_amd import build as B

def main():
    src = sys.argv[1]
    B.build()
    objs = [os.path.join(B.OBJDIR, s.replace(".hip", ".o")) for s in B.SOURCES]
    def one(spec):
        name, flags = spec.split("=", 1)
        obj = os.path.join(B.OBJDIR, "abl_%s_%s" % (name, src.replace(".hip", ".o")))
        cmd = [B._hipcc(), *B.FLAGS, *[(f[1:] if f.startswith("@") else "-D" + f) for f in flags.split(",") if f], "-c", os.path.join(B.CSRC, src), "-o", obj]
        subprocess.run(cmd, check=True)
        lib = os.path.join(B.LIBDIR, "libabl_%s.so" % name)
        link = [o if os.path.basename(o) != src.replace(".hip", ".o") else obj for o in objs]
        subprocess.run([B._hipcc(), "-shared", "-fPIC", "--offload-arch=" + B.ARCH, *link, "-o", lib], check=True)
        return lib
    with ThreadPoolExecutor(max_workers=4) as ex:
        for lib in ex.map(one, sys.argv[2:]):
            print(lib)

if __name__ == "__main__":
    main()
