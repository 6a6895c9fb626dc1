// armour_worker -- the process that owns the GPU for the file-protocol executables: runs one planning iteration of
// either protocol, or stays resident (`--serve`).  armour_main / armtd_main start it when no resident planner answers
// (planner_client.cpp); see cli_common.h.
#include "cli_common.h"

int main(int argc, char** argv) { return cli::worker_main(argc, argv); }
