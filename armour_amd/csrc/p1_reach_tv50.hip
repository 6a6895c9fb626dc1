// Second instantiation of the time-vectorised reach-set kernel (p1_tv.inc.h on the arithmetic of pz_tv.h): the same source with ROWS OF 50 DOUBLES.
// BASELINE's trajectories have T = 100 time intervals (RT/Parameters.h:16 has 128): two groups of 50 time steps per problem, so with the 64-double
// rows of p1_reach.hip every fourth 128-byte line of every row was moved for two lanes.  Packed rows of 400 bytes cut the L2-miss traffic of a
// 128-problem build from 28.2 to 22.8 GB.  The row width is a compile-time constant of the kernel (every row offset is an
// immediate of its load), hence a translation unit of its own; armour_p1_build (p1_reach.hip) picks the unit by the group size.
#define P1_TV_VARIANT
#ifndef TV_GROW
#define TV_GROW 50
#endif
#define P1_TV_LAUNCH armour_p1_tv_launch_g50
#include "p1_reach.hip"
