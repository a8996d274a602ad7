#!/usr/bin/env python3
"""Applies the edits of INTEGRATION.md sections 2-3 to SCRATCH copies of three reference files, so that the
real dispatcher arm -- RayTrace::create_image(info, "hip" | "hip-multigpu" | "auto") -- and the real
CreateImage harness can be compiled with -DUSE_HIP and run on the GPU box (oracle/Makefile `dispatcher`).

  apply_integration.py <reference/src> <scratch dir>

The reference tree is read-only and its sources never enter this repository: the scratch directory lives
under oracle/_ref/ (git-ignored), is compiled at once and removed by the Makefile; only the binary
remains.  Every edit is an insertion next to an anchor line of the reference file; a missing anchor is an
error (the reference changed: update INTEGRATION.md and this script together)."""
import sys
from pathlib import Path

SIG = ("int N, const RayTrace::EUV_beam_struct& euv_beam, const RayTrace::ray_gain_struct *gain,\n"
       "    const RayTrace::ray_seed_struct *seed, int method, const std::vector<ray_struct> &rays,\n"
       "    double scale, double *image, double *I_ang, unsigned int &failure_code,\n"
       "    std::vector<ray_struct> &failed_rays")

EXTERN = f"""#if defined( USE_HIP )
extern void RayTraceImageHipLoop( {SIG} );
extern void RayTraceImageHipMultiGPULoop( {SIG} );
#endif
"""

ARMS = """    } else if ( compute_method == "hip" ) {
#if defined( USE_HIP )
        RayTraceImageHipLoop( N, std::ref(*info->euv_beam), info->gain, info->seed,
            method, rays, scale, image, I_ang, failure_code, failed_rays );
#else
        RAY_ERROR( "Hip is not availible" );
#endif
    } else if ( compute_method == "hip-multigpu" ) {
#if defined( USE_HIP )
        RayTraceImageHipMultiGPULoop( N, std::ref(*info->euv_beam), info->gain, info->seed,
            method, rays, scale, image, I_ang, failure_code, failed_rays );
#else
        RAY_ERROR( "Hip-MultiGPU is not availible" );
#endif
"""


def insert_before(text: str, anchor: str, new: str, what: str) -> str:
    i = text.find(anchor)
    if i < 0:
        raise SystemExit(f"apply_integration: anchor not found ({what})")
    return text[:i] + new + text[i:]


def replace_once(text: str, old: str, new: str, what: str) -> str:
    if text.count(old) < 1:
        raise SystemExit(f"apply_integration: anchor not found ({what})")
    return text.replace(old, new, 1)


def main() -> None:
    src, out = Path(sys.argv[1]), Path(sys.argv[2])
    out.mkdir(parents=True, exist_ok=True)
    # ---- src/RayTraceImage.cpp: extern declarations, "auto" precedence, two dispatch arms
    t = (src / "RayTraceImage.cpp").read_text()
    t = insert_before(t, "/**********************************************************************\n* Call RayTraceImage function from a thread loop",
                      EXTERN + "\n\n", "extern declarations")
    t = replace_once(t, '#if defined( ENABLE_OPENACC )\n        compute_method = "openacc";',
                     '#if defined( USE_HIP )\n        compute_method = "hip";\n#elif defined( ENABLE_OPENACC )\n        compute_method = "openacc";',
                     "auto precedence")
    t = insert_before(t, '    } else if ( compute_method == "cpu" ) {', ARMS, "dispatch chain")
    (out / "RayTraceImage.cpp").write_text(t)
    # ---- src/CreateImage.cpp: default method list, warm-up search
    t = (src / "CreateImage.cpp").read_text()
    t = replace_once(t, '#ifdef USE_OPENACC\n        methods.push_back( "OpenAcc" );\n#endif\n',
                     '#ifdef USE_OPENACC\n        methods.push_back( "OpenAcc" );\n#endif\n'
                     '#ifdef USE_HIP\n        methods.push_back( "Hip" );\n        methods.push_back( "Hip-MultiGPU" );\n#endif\n',
                     "default method list")
    t = replace_once(t, 'auto index = std::find(methods.begin(),methods.end(),"Cuda-MultiGPU");',
                     'auto index = std::find(methods.begin(),methods.end(),"Hip-MultiGPU");\n'
                     '        if ( index == methods.end() )\n'
                     '            index = std::find(methods.begin(),methods.end(),"Hip");\n'
                     '        if ( index == methods.end() )\n'
                     '            index = std::find(methods.begin(),methods.end(),"Cuda-MultiGPU");',
                     "warm-up search")
    (out / "CreateImage.cpp").write_text(t)
    # ---- src/CreateImageHelpers.h: help text
    t = (src / "CreateImageHelpers.h").read_text()
    t = replace_once(t, "cpu, threads, OpenMP, Cuda, Cuda-MultiGPU, OpenAcc,", "cpu, threads, OpenMP, Hip, Hip-MultiGPU, Cuda, Cuda-MultiGPU, OpenAcc,",
                     "help text")
    (out / "CreateImageHelpers.h").write_text(t)


if __name__ == "__main__":
    main()
