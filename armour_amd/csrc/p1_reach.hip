// P1: reach-set build on the device (placeholder until the JRS / FK / RNEA kernels land).
#include "common.h"

int armour_p1_build(ArmourPlanner* h, const double* obstacles) {
    (void)h; (void)obstacles;
    armour_set_error("armour_set_problems: device reach-set build not available in this build");
    return ARMOUR_ESTATE;
}
void armour_p1_free(ArmourPlanner* h) { (void)h; }
