/*
 * tests/stubs/mex_harness.cpp -- TEST INFRASTRUCTURE.  Plays MATLAB for armour_amd/mex/armour_hip_mex.cpp: builds the
 * prhs[] of one call from plain C arrays, runs mexFunction, keeps the plhs[] for the caller to copy out.  ctypes-friendly
 * (tests/test_mex_gateway.py).  Arguments are real double matrices (column-major) or strings.
 */
#include <vector>

#include "mex.h"

extern "C" {
int mexstub_locked = 0;
void (*mexstub_at_exit)(void) = nullptr;
}

namespace {
std::vector<mxArray*> g_out;
std::string g_err;
void drop_outputs() {
    for (mxArray* a : g_out) mxDestroyArray(a);
    g_out.clear();
}
}  // namespace

extern "C" {

/* One call  [out1, ..., out_nlhs] = mexfile(args...).  Argument i is the string strs[i] when strs[i] != NULL, else the
 * rows[i] x cols[i] double matrix data[i].  Returns the number of outputs held (>= 1 if the gateway assigned plhs[0]),
 * or -1 after mexErrMsgTxt (message: mexh_error()). */
int mexh_call(int nlhs, int nrhs, const char* const* strs, const double* const* data, const int* rows, const int* cols) {
    drop_outputs();
    g_err.clear();
    std::vector<mxArray*> in(nrhs, nullptr);
    for (int i = 0; i < nrhs; i++) {
        if (strs[i]) in[i] = mxCreateString(strs[i]);
        else {
            in[i] = mxCreateDoubleMatrix((mwSize)rows[i], (mwSize)cols[i], mxREAL);
            if (rows[i] * cols[i] > 0) memcpy(mxGetPr(in[i]), data[i], sizeof(double) * (size_t)rows[i] * (size_t)cols[i]);
        }
    }
    const int nout = nlhs > 1 ? nlhs : 1;   /* MATLAB always provides room for plhs[0] (ans) */
    std::vector<mxArray*> out(nout, nullptr);
    int rc = 0;
    try {
        mexFunction(nlhs, out.data(), nrhs, const_cast<const mxArray**>(in.data()));
        for (int i = 0; i < nout; i++)
            if (out[i]) { g_out.resize(i + 1, nullptr); g_out[i] = out[i]; }
        rc = (int)g_out.size();
    } catch (const MexError& e) {
        g_err = e.msg;
        for (mxArray* a : out) mxDestroyArray(a);
        rc = -1;
    }
    for (mxArray* a : in) mxDestroyArray(a);
    return rc;
}
const char* mexh_error(void) { return g_err.c_str(); }
int mexh_out_dims(int i, int* rows, int* cols, int* is_logical) {
    if (i < 0 || i >= (int)g_out.size() || !g_out[i]) return -1;
    *rows = (int)g_out[i]->m; *cols = (int)g_out[i]->n; *is_logical = g_out[i]->cls == mxLOGICAL_CLASS;
    return 0;
}
int mexh_out_copy(int i, double* dst) {
    if (i < 0 || i >= (int)g_out.size() || !g_out[i]) return -1;
    const mxArray* a = g_out[i];
    for (size_t k = 0; k < a->m * a->n; k++) dst[k] = a->cls == mxLOGICAL_CLASS ? (double)a->lg[k] : a->pr[k];
    return 0;
}
int mexh_is_locked(void) { return mexstub_locked; }
/* what MATLAB does at `clear mex` / exit */
void mexh_exit(void) {
    drop_outputs();
    if (mexstub_at_exit) mexstub_at_exit();
    mexstub_at_exit = nullptr;
    mexstub_locked = 0;
}
}
