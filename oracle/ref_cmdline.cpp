// oracle/ref_cmdline.cpp — TEST INFRASTRUCTURE ONLY (built into oracle/_ref/, only where /root/reference is mounted).
//
// Drives the reference's OWN command-line parser — Utility::CmdLine, CmdLine.h + CmdLine.cpp, compiled unmodified from where they lie
// (oracle/Makefile target _ref; nothing of the reference is copied into this repository) — with the parameter registrations of
// get_input (main.cu:29-44) on a configuration that carries Config::Config()'s defaults (ColorTransfer/Config.h:58-72; Config.h itself
// is not included: it pulls in the colour tables and OpenCV types). After Parse it prints what main() would go on with.
// The output pins D1 of SURVEY §8(a): tests/golden/gen_cmdline_ref.py records it for a set of argument vectors
// (tests/golden/cmdline_ref.json), tests/test_cli.py::test_cli_parser_matches_the_reference_cmdline runs the product CLI's
// `--parse-only` hook on the same vectors.
//
// usage: ref_cmdline <args of neural_color_transfer.exe ...>
//   stdout: whatever Parse prints (help / "Unrecognized parameter"), then "@@RESULT rc=<1 parsed | 0 main returns -1>" and one "key=value" line per parameter
#include <cstdio>
#include <string>
#include "CmdLine.h"

int main(int argc, char** argv) {
    Utility::CmdLine cmdLine;
    std::string modelDir, inputDir, outputDir;                       // Config.h:82-84 (empty by default)
    int gpuId = 0;                                                   // main.cu:551
    double reverseWeight = 2.0, varEpslon = 0.60, nonlocalWeight = 2.0, localWeight = 0.125, wlsLamdaInit = 0.024;   // Config.h:61-65
    cmdLine.Param("m", modelDir, "Directory of network models.");                                                 // main.cu:31
    cmdLine.Param("i", inputDir, "Input directory of content and style images and pairs.txt.");                   // :32
    cmdLine.Param("o", outputDir, "Output directory of result images.");                                          // :33
    cmdLine.Param("g", gpuId, "GPU ID (default: 0).");                                                            // :34
    cmdLine.Param("bds", reverseWeight, "Weight of reverse color in BDS voting (default: 2.0).");                 // :37
    cmdLine.Param("eps", varEpslon, "Eps is used to avoid dividing zero (default: 0.6 with range in [0-255]).");   // :40
    cmdLine.Param("nl", nonlocalWeight, "Weight of nonlocal constraint (default: 0.4.");                          // :41
    cmdLine.Param("l", localWeight, "Weight of local constraitn (default: 0.001).");                              // :42
    cmdLine.Param("w", wlsLamdaInit, "Initial value of WLS weight (default: 0.0234375).");                        // :43
    const bool ok = cmdLine.Parse(argc, argv);
    fflush(stdout);
    std::cout << std::flush;
    printf("@@RESULT rc=%d\n", ok ? 1 : 0);
    printf("m=%s\ni=%s\no=%s\ng=%d\n", modelDir.c_str(), inputDir.c_str(), outputDir.c_str(), gpuId);
    printf("bds=%.17g\neps=%.17g\nnl=%.17g\nl=%.17g\nw=%.17g\n", reverseWeight, varEpslon, nonlocalWeight, localWeight, wlsLamdaInit);
    printf("files=%d\n", cmdLine.NumFiles());
    return 0;
}
