/*
 * glbopts.h — the compile-time constants and allocator macros the reference's CPython glue takes from the SCS
 * core's header of this name (`#include "glbopts.h"`, R:scs/scspy.c:16).  The upstream file (scs_source/include/
 * glbopts.h) is ABSENT from the reference snapshot (R:.gitmodules:1-3); what is declared here is exactly what the
 * glue uses, each with the line that uses it.  With this header, scs.h and scs_types.h on the include path,
 * R:scs/scspy.c compiles and links against libscs_hip.so unchanged (tests/test_reference_glue_cpu.py builds it).
 */
#ifndef SCS_GLBOPTS_H_GUARD
#define SCS_GLBOPTS_H_GUARD

#include <stdlib.h>
#include <stdio.h>
#include <math.h> /* isfinite(): R:scs/scsobject.h:805-866 (settings range checks) */

#include "scs_types.h"

/* Defaults of the boolean settings the glue fills in when the caller passes None:
 * R:scs/scsobject.h:796-800 (verbose, normalize, adaptive_scale) and :869 (warm_start).  Values = the defaults of
 * SCS 3.2.x that scs_set_default_settings() of this library writes (csrc/scs_hip.hip) and that the front end documents
 * (R:scs/py/__init__.py solve() docstring; R:test/test_scs_coverage.py exercises verbose=False explicitly). */
#define VERBOSE (1)
#define NORMALIZE (1)
#define ADAPTIVE_SCALE (1)
#define WARM_START (0)

/* Allocator macros: R:scs/scsobject.h:161,171,179 (cone arrays), :455-461 (data carriers), :875-890 (solution). */
#define scs_malloc malloc
#define scs_calloc calloc
#define scs_realloc realloc
#define scs_free free

#define SCS_NULL 0

#endif
