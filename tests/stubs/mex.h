/*
 * tests/stubs/mex.h -- TEST INFRASTRUCTURE.  A minimal, FUNCTIONAL stand-in for MATLAB's MEX C API, just large enough to
 * compile armour_amd/mex/armour_hip_mex.cpp with g++ and to drive its mexFunction from a test harness
 * (tests/stubs/mex_harness.cpp).  MATLAB is not in the build image, so without this the gateway would never meet a
 * compiler.  Signatures follow the documented C Matrix / MEX API (real double matrices, column-major); nothing here is
 * used by the product or to build the reference.  mexErrMsgTxt throws MexError where MATLAB unwinds to the prompt.
 */
#ifndef ARMOUR_TEST_MEX_STUB_H
#define ARMOUR_TEST_MEX_STUB_H

#include <stddef.h>
#include <stdlib.h>
#include <string.h>

#include <string>

typedef size_t mwSize;
typedef size_t mwIndex;
typedef bool mxLogical;
typedef enum { mxUNKNOWN_CLASS = 0, mxLOGICAL_CLASS = 3, mxCHAR_CLASS = 4, mxDOUBLE_CLASS = 6 } mxClassID;
typedef enum { mxREAL = 0, mxCOMPLEX = 1 } mxComplexity;

struct mxArray_tag {
    mxClassID cls;
    mwSize m, n;
    double* pr;        /* mxDOUBLE_CLASS data, column-major */
    mxLogical* lg;     /* mxLOGICAL_CLASS data */
    char* str;         /* mxCHAR_CLASS: NUL-terminated row vector */
};
typedef struct mxArray_tag mxArray;

struct MexError { std::string msg; };

static inline mxArray* mxstub_new(mxClassID cls, mwSize m, mwSize n) {
    mxArray* a = (mxArray*)calloc(1, sizeof(mxArray));
    a->cls = cls; a->m = m; a->n = n;
    if (cls == mxDOUBLE_CLASS) a->pr = (double*)calloc(m * n > 0 ? m * n : 1, sizeof(double));
    if (cls == mxLOGICAL_CLASS) a->lg = (mxLogical*)calloc(m * n > 0 ? m * n : 1, sizeof(mxLogical));
    return a;
}
static inline mxArray* mxCreateNumericMatrix(mwSize m, mwSize n, mxClassID cls, mxComplexity) { return mxstub_new(cls, m, n); }
static inline mxArray* mxCreateDoubleMatrix(mwSize m, mwSize n, mxComplexity) { return mxstub_new(mxDOUBLE_CLASS, m, n); }
static inline mxArray* mxCreateDoubleScalar(double v) { mxArray* a = mxstub_new(mxDOUBLE_CLASS, 1, 1); a->pr[0] = v; return a; }
static inline mxArray* mxCreateLogicalScalar(mxLogical v) { mxArray* a = mxstub_new(mxLOGICAL_CLASS, 1, 1); a->lg[0] = v; return a; }
static inline mxArray* mxCreateString(const char* s) {
    mxArray* a = mxstub_new(mxCHAR_CLASS, 1, strlen(s));
    a->str = strdup(s);
    return a;
}
static inline void mxDestroyArray(mxArray* a) {
    if (!a) return;
    free(a->pr); free(a->lg); free(a->str); free(a);
}
static inline double* mxGetPr(const mxArray* a) { return a->pr; }
static inline void* mxGetData(const mxArray* a) { return a->cls == mxLOGICAL_CLASS ? (void*)a->lg : (void*)a->pr; }
static inline mwSize mxGetM(const mxArray* a) { return a->m; }
static inline mwSize mxGetN(const mxArray* a) { return a->n; }
static inline size_t mxGetNumberOfElements(const mxArray* a) { return a->m * a->n; }
static inline double mxGetScalar(const mxArray* a) { return a->cls == mxLOGICAL_CLASS ? (double)a->lg[0] : a->pr[0]; }
static inline bool mxIsDouble(const mxArray* a) { return a->cls == mxDOUBLE_CLASS; }
static inline bool mxIsChar(const mxArray* a) { return a->cls == mxCHAR_CLASS; }
/* 0 on success, 1 if the array is not a string or does not fit (as documented) */
static inline int mxGetString(const mxArray* a, char* buf, mwSize buflen) {
    if (!a || a->cls != mxCHAR_CLASS || !a->str || strlen(a->str) + 1 > buflen) return 1;
    strcpy(buf, a->str);
    return 0;
}
static inline void* mxMalloc(size_t n) { return malloc(n ? n : 1); }
static inline void* mxCalloc(size_t n, size_t sz) { return calloc(n ? n : 1, sz ? sz : 1); }
static inline void mxFree(void* p) { free(p); }

/* MEX-file state, owned by the harness */
extern "C" {
extern int mexstub_locked;
extern void (*mexstub_at_exit)(void);
}
static inline void mexLock(void) { mexstub_locked++; }
static inline void mexUnlock(void) { if (mexstub_locked > 0) mexstub_locked--; }
static inline bool mexIsLocked(void) { return mexstub_locked > 0; }
static inline int mexAtExit(void (*fn)(void)) { mexstub_at_exit = fn; return 0; }
[[noreturn]] static inline void mexErrMsgTxt(const char* msg) { throw MexError{msg ? msg : ""}; }
[[noreturn]] static inline void mexErrMsgIdAndTxt(const char*, const char* msg, ...) { throw MexError{msg ? msg : ""}; }

extern "C" void mexFunction(int nlhs, mxArray* plhs[], int nrhs, const mxArray* prhs[]);

#endif
