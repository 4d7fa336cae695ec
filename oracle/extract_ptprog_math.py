#!/usr/bin/env python3
"""Build step of oracle/_ref (TEST INFRASTRUCTURE, NOT PRODUCT).

PathTracer_Optix/pathTracerPrograms.cu cannot be compiled here as a whole: it includes <optix.h>
(OptiX SDK, absent) and we write no stand-in headers.  But the sampling / BSDF helpers inside it
touch no OptiX symbol at all — they need only sutil/vec_math.h and libm:

    OrthonormalBasis              :54-85
    safeDivide (float, float3)    :265-268, :281-284
    cosine_sample_hemisphere      :341-353
    uniform_sample_hemisphere     :368-380
    sampleGGX                     :455-476
    fresnelSchlickConductor       :494-510
    FrDielectric                  :534-559

This script copies exactly those definitions, verbatim, from where the file lies under
/root/reference into oracle/_ref/ptprog_math.inc at BUILD time (oracle/_ref/ is git-ignored and
gpurun-ignored: nothing extracted is committed or shipped).  ref_math_shim.cpp includes the result
after <sutil/vec_math.h>, so libref.so then runs the reference's own text of these functions.

Each definition is located by its signature and must start inside the line window the citations
above give (a few lines of slack); anything else fails the build loudly rather than silently
pinning against the wrong text.
"""
import os
import re
import sys

SRC = "/root/reference/PathTracer_Optix/pathTracerPrograms.cu"

# (name, regex of the first line, expected first line, expected last line)
WANTED = [
    ("OrthonormalBasis", r"^struct OrthonormalBasis\s*$", 54, 85),
    ("safeDivide(float)", r"^static __forceinline__ __device__ float safeDivide\(float a, float b\)\s*$", 265, 268),
    ("safeDivide(float3)", r"^static __forceinline__ __device__ float3 safeDivide\(float3 a, float b\)\s*\{\s*$", 281, 284),
    ("cosine_sample_hemisphere", r"^static __forceinline__ __device__ void cosine_sample_hemisphere\(const float eta1, const float eta2, float3& p\)\s*$", 341, 353),
    ("uniform_sample_hemisphere", r"^static __forceinline__ __device__ void uniform_sample_hemisphere\(const float u1, const float u2, float3& wi\)\s*$", 368, 380),
    ("sampleGGX", r"^static __forceinline__ __device__ float3 sampleGGX\(float u1, float u2, float roughness, const float3& N\)\s*$", 455, 476),
    ("fresnelSchlickConductor", r"^static __forceinline__ __device__ float3 fresnelSchlickConductor\(float cosTheta, float3 eta, float3 k\)\s*$", 494, 510),
    ("FrDielectric", r"^static __forceinline__ __device__ float FrDielectric\(float cosThetaI, float etaI, float etaT\)\s*\{\s*$", 534, 559),
]
SLACK = 3
FORBIDDEN = re.compile(r"optix|params\.|__constant__", re.I)


def die(msg):
    sys.stderr.write("extract_ptprog_math: " + msg + "\n")
    sys.exit(1)


def main():
    out_path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.abspath(__file__)), "_ref", "ptprog_math.inc")
    if not os.path.exists(SRC):
        die("reference source not found: " + SRC)
    with open(SRC, "r", encoding="utf-8", errors="strict") as f:
        lines = f.read().split("\n")
    chunks = []
    for name, pat, first, last in WANTED:
        rx = re.compile(pat)
        hits = [i + 1 for i, ln in enumerate(lines) if rx.match(ln)]
        if len(hits) != 1:
            die("%s: signature found %d times (expected once)" % (name, len(hits)))
        start = hits[0]
        if abs(start - first) > SLACK:
            die("%s: signature at line %d, expected near %d" % (name, start, first))
        # definition ends at the first closing brace in column 0 ('}' or '};')
        end = None
        for j in range(start, len(lines)):
            if re.match(r"^\};?\s*$", lines[j]):
                end = j + 1
                break
        if end is None or abs(end - last) > SLACK:
            die("%s: closing brace at line %s, expected near %d" % (name, end, last))
        body = lines[start - 1:end]
        code_only = "\n".join(re.sub(r"//.*$", "", ln) for ln in body)
        if FORBIDDEN.search(code_only):
            die("%s: body refers to OptiX / launch state; it is not OptiX-free" % name)
        chunks.append("/* ---- %s: pathTracerPrograms.cu:%d-%d, verbatim ---- */\n%s\n" % (name, start, end, "\n".join(body)))
    os.makedirs(os.path.dirname(out_path), exist_ok=True)
    with open(out_path, "w", encoding="utf-8") as f:
        f.write("/* GENERATED at build time by oracle/extract_ptprog_math.py from %s.\n"
                "   Never committed, never shipped (oracle/_ref/ is git- and gpurun-ignored). */\n\n" % SRC)
        f.write("\n".join(chunks))
    print("extract_ptprog_math: wrote %s (%d definitions)" % (out_path, len(chunks)))


if __name__ == "__main__":
    main()
