"""Build the REAL reference kernel (`remixt/bpmodel.pyx`) into oracle/_ref/.

TEST INFRASTRUCTURE ONLY.  Nothing here is shipped or measured as the product.

The reference's hot path is a single Cython file that only needs libc + numpy.
This recipe cythonizes it *where it lies* under /root/reference (no source is
copied into this repository) and writes every output -- the generated C++, the
object files and the extension module -- under oracle/_ref/, which is
git-ignored.  The generated C++ (which embeds the reference text as comments)
is deleted after the build so only the binary module remains.

Result:  oracle/_ref/remixt/bpmodel.<abi>.so   (import as `remixt.bpmodel` via
oracle/refload.py, which assembles a `remixt` namespace without touching the
reference tree).

Usage:  python oracle/build_ref.py [--force]
"""
import glob
import os
import shutil
import sys
import sysconfig

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.environ.get("REMIXT_REFERENCE", "/root/reference")
OUT = os.path.join(HERE, "_ref")


def ref_module_path():
    suffix = sysconfig.get_config_var("EXT_SUFFIX")
    return os.path.join(OUT, "remixt", "bpmodel" + suffix)


def build(force=False):
    pyx = os.path.join(REF, "remixt", "bpmodel.pyx")
    target = ref_module_path()
    if os.path.exists(target) and not force:
        return target
    if not os.path.exists(pyx):
        return None  # reference not mounted (e.g. on the GPU box): use prebuilt or nothing

    import numpy
    from Cython.Build import cythonize
    from setuptools import Extension
    from setuptools.dist import Distribution
    from setuptools.command.build_ext import build_ext

    build_dir = os.path.join(OUT, "build")
    os.makedirs(os.path.join(OUT, "remixt"), exist_ok=True)
    os.makedirs(build_dir, exist_ok=True)

    ext = Extension(
        "remixt.bpmodel", [pyx],
        include_dirs=[numpy.get_include()],
        language="c++",
        extra_compile_args=["-O2", "-w"],
        define_macros=[("NPY_NO_DEPRECATED_API", "NPY_1_7_API_VERSION")],
    )
    # build_dir receives the generated .cpp; nothing is written next to the .pyx
    exts = cythonize([ext], build_dir=build_dir, quiet=True,
                     compiler_directives={"language_level": 3})
    dist = Distribution({"name": "remixt_ref", "ext_modules": exts})
    cmd = build_ext(dist)
    cmd.build_lib = OUT
    cmd.build_temp = build_dir
    cmd.inplace = False
    cmd.ensure_finalized()
    cmd.run()
    # drop generated C++ / objects: only the binary module stays
    shutil.rmtree(build_dir, ignore_errors=True)
    for stray in glob.glob(os.path.join(OUT, "**", "*.cpp"), recursive=True):
        os.remove(stray)
    assert os.path.exists(target), target
    return target


if __name__ == "__main__":
    p = build(force="--force" in sys.argv)
    print(p if p else "reference not present; nothing built")
