/*
 * scs.h — stands in for the SCS core's public header of this name (`#include "scs.h"`, R:scs/scspy.c:18; upstream
 * scs_source/include/scs.h is absent from the reference snapshot).  Everything the glue needs from it — the data
 * carriers and the six entry points scs_init / scs_solve / scs_update / scs_finish / scs_set_default_settings /
 * scs_version (R:scs/scsobject.h:520,903,986,1217,1240, R:scs/scsmodule.h:5) — is declared by scs_hip.h PART 1.
 */
#ifndef SCS_H_GUARD
#define SCS_H_GUARD
#include "scs_hip.h"
#endif
