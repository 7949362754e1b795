"""Build the REFERENCE's Cython wrapper (particlefilter/cython/particlefilter_wrapper.pyx, unchanged) against
libparticlefilter.so instead of the reference's particlefilter/src/*.c -- the one Extension(...) call of INTEGRATION.md
section 2 -- into oracle/_ref/cy/ (git-ignored like everything under oracle/_ref; it travels to the GPU box, where
tests/test_gpu_dropin.py drives the HIP library through it).  Build container only: needs /root/reference and Cython.
The .pyx is copied to a scratch directory because Cython writes its C file next to it and the reference tree is read-only;
nothing of it enters the repository."""
import os
import shutil
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
PYX = "/root/reference/particlefilter/cython/particlefilter_wrapper.pyx"
OUT = os.path.join(HERE, "_ref", "cy")


def main():
    if not os.path.isfile(PYX):
        print("build_ref_cython: no reference tree here, nothing to do")
        return 0
    try:
        import Cython  # noqa: F401
    except ImportError:
        print("build_ref_cython: Cython is not importable, skipped")
        return 0
    inc = os.path.join(ROOT, "include")
    libdir = os.path.join(ROOT, "slam3d_amd", "lib")
    work = tempfile.mkdtemp(prefix="pfcy_")
    try:
        os.makedirs(os.path.join(work, "particlefilter", "cython"))
        shutil.copy(PYX, os.path.join(work, "particlefilter", "cython"))
        with open(os.path.join(work, "setup.py"), "w") as fh:
            fh.write(
                "import numpy\nfrom setuptools import setup, Extension\nfrom Cython.Build import cythonize\n"
                "ext = Extension('particlefilter',\n"
                "                sources=['particlefilter/cython/particlefilter_wrapper.pyx'],\n"
                f"                include_dirs=[{inc!r}, numpy.get_include()],\n"
                f"                libraries=['particlefilter'], library_dirs=[{libdir!r}],\n"
                "                extra_link_args=['-Wl,-rpath,$ORIGIN/../../../slam3d_amd/lib'])\n"
                "setup(name='particlefilter', ext_modules=cythonize([ext], language_level=3))\n")
        r = subprocess.run([sys.executable, "setup.py", "-q", "build_ext", "--inplace"], cwd=work, capture_output=True, text=True)
        if r.returncode != 0:
            sys.stderr.write(r.stdout[-3000:] + r.stderr[-3000:])
            return 1
        built = [f for f in os.listdir(work) if f.startswith("particlefilter") and f.endswith(".so")]
        if not built:
            sys.stderr.write("build_ref_cython: no module was built\n")
            return 1
        os.makedirs(OUT, exist_ok=True)
        for f in os.listdir(OUT):
            os.remove(os.path.join(OUT, f))
        shutil.copy(os.path.join(work, built[0]), os.path.join(OUT, built[0]))
        print("build_ref_cython:", os.path.join(OUT, built[0]))
        return 0
    finally:
        shutil.rmtree(work, ignore_errors=True)


if __name__ == "__main__":
    sys.exit(main())
