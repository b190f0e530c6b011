/*
 * mex_stub.c -- host-memory implementation of the mx and mex functions declared in mex/stub/mex.h, plus the harness entry
 * rmxstub_call() through which tests/test_mex_gateway.py runs mexFunction outside MATLAB.  Test infrastructure only.
 * Column-major storage, like MATLAB.  A struct array is 1x1 here (all the gateway needs).
 */
#define _POSIX_C_SOURCE 200809L
#include <setjmp.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "mex.h"

#define MAXDIMS 4
#define MAXFIELDS 48

struct mxArray_tag {
    mxClassID cls;
    mwSize ndim;
    mwSize dims[MAXDIMS];
    size_t elsize;
    void* data;               /* numeric / char data */
    int nfields;              /* struct */
    char* names[MAXFIELDS];
    mxArray* values[MAXFIELDS];
};

static size_t class_size(mxClassID c) {
    switch (c) {
        case mxDOUBLE_CLASS: case mxINT64_CLASS: case mxUINT64_CLASS: return 8;
        case mxSINGLE_CLASS: case mxINT32_CLASS: case mxUINT32_CLASS: return 4;
        case mxINT16_CLASS: case mxUINT16_CLASS: case mxCHAR_CLASS: return 2;
        case mxINT8_CLASS: case mxUINT8_CLASS: case mxLOGICAL_CLASS: return 1;
        default: return 0;
    }
}
size_t mxGetNumberOfElements(const mxArray* a) {
    size_t n = 1;
    for (mwSize i = 0; i < a->ndim; ++i) n *= a->dims[i];
    return n;
}
mxArray* mxCreateNumericArray(mwSize ndim, const mwSize* dims, mxClassID cls, mxComplexity flag) {
    (void)flag;
    if (ndim > MAXDIMS) return NULL;
    mxArray* a = (mxArray*)calloc(1, sizeof *a);
    a->cls = cls;
    a->ndim = ndim < 2 ? 2 : ndim;
    a->dims[0] = a->dims[1] = 1;
    for (mwSize i = 0; i < ndim; ++i) a->dims[i] = dims[i];
    a->elsize = class_size(cls);
    const size_t n = mxGetNumberOfElements(a);
    a->data = calloc(n ? n : 1, a->elsize ? a->elsize : 1);
    return a;
}
mxArray* mxCreateNumericMatrix(mwSize m, mwSize n, mxClassID cls, mxComplexity flag) {
    const mwSize d[2] = {m, n};
    return mxCreateNumericArray(2, d, cls, flag);
}
mxArray* mxCreateDoubleMatrix(mwSize m, mwSize n, mxComplexity flag) { return mxCreateNumericMatrix(m, n, mxDOUBLE_CLASS, flag); }
mxArray* mxCreateDoubleScalar(double v) {
    mxArray* a = mxCreateDoubleMatrix(1, 1, mxREAL);
    *(double*)a->data = v;
    return a;
}
mxArray* mxCreateString(const char* s) {
    const size_t n = strlen(s);
    mxArray* a = mxCreateNumericMatrix(1, n, mxCHAR_CLASS, mxREAL);
    for (size_t i = 0; i < n; ++i) ((uint16_t*)a->data)[i] = (uint16_t)(unsigned char)s[i];
    return a;
}
mxArray* mxCreateStructMatrix(mwSize m, mwSize n, int nfields, const char** names) {
    if (m != 1 || n != 1 || nfields > MAXFIELDS) return NULL;
    mxArray* a = (mxArray*)calloc(1, sizeof *a);
    a->cls = mxSTRUCT_CLASS;
    a->ndim = 2;
    a->dims[0] = a->dims[1] = 1;
    a->nfields = nfields;
    for (int i = 0; i < nfields; ++i) a->names[i] = strdup(names[i]);
    return a;
}
void mxDestroyArray(mxArray* a) {
    if (!a) return;
    for (int i = 0; i < a->nfields; ++i) {
        free(a->names[i]);
        mxDestroyArray(a->values[i]);
    }
    free(a->data);
    free(a);
}
double* mxGetPr(const mxArray* a) { return (double*)a->data; }
void* mxGetData(const mxArray* a) { return a->data; }
size_t mxGetM(const mxArray* a) { return a->dims[0]; }
size_t mxGetN(const mxArray* a) { return mxGetNumberOfElements(a) / (a->dims[0] ? a->dims[0] : 1); }
double mxGetScalar(const mxArray* a) {
    if (!a->data || mxGetNumberOfElements(a) == 0) return 0.0;
    switch (a->cls) {
        case mxDOUBLE_CLASS: return *(double*)a->data;
        case mxSINGLE_CLASS: return *(float*)a->data;
        case mxINT32_CLASS: return *(int32_t*)a->data;
        case mxUINT32_CLASS: return *(uint32_t*)a->data;
        case mxINT64_CLASS: return (double)*(int64_t*)a->data;
        case mxUINT64_CLASS: return (double)*(uint64_t*)a->data;
        case mxLOGICAL_CLASS: case mxUINT8_CLASS: return *(uint8_t*)a->data;
        default: return 0.0;
    }
}
int mxGetString(const mxArray* a, char* buf, mwSize buflen) {
    if (a->cls != mxCHAR_CLASS) return 1;
    const size_t n = mxGetNumberOfElements(a);
    if (n + 1 > buflen) return 1;
    for (size_t i = 0; i < n; ++i) buf[i] = (char)((uint16_t*)a->data)[i];
    buf[n] = 0;
    return 0;
}
mxArray* mxGetField(const mxArray* a, mwIndex index, const char* name) {
    if (a->cls != mxSTRUCT_CLASS || index != 0) return NULL;
    for (int i = 0; i < a->nfields; ++i)
        if (!strcmp(a->names[i], name)) return a->values[i];
    return NULL;
}
void mxSetField(mxArray* a, mwIndex index, const char* name, mxArray* v) {
    if (a->cls != mxSTRUCT_CLASS || index != 0) return;
    for (int i = 0; i < a->nfields; ++i)
        if (!strcmp(a->names[i], name)) {
            a->values[i] = v;
            return;
        }
}
int mxIsStruct(const mxArray* a) { return a->cls == mxSTRUCT_CLASS; }
int mxIsDouble(const mxArray* a) { return a->cls == mxDOUBLE_CLASS; }
int mxIsInt32(const mxArray* a) { return a->cls == mxINT32_CLASS; }
int mxIsUint64(const mxArray* a) { return a->cls == mxUINT64_CLASS; }
int mxIsComplex(const mxArray* a) { (void)a; return 0; }
int mxIsEmpty(const mxArray* a) { return mxGetNumberOfElements(a) == 0; }
void* mxCalloc(size_t n, size_t size) { return calloc(n, size); }
void mxFree(void* p) { free(p); }
void mexMakeMemoryPersistent(void* p) { (void)p; }
static void (*g_exit_fcn)(void) = NULL;
int mexAtExit(void (*f)(void)) { g_exit_fcn = f; return 0; }   /* MATLAB calls it on `clear mex`; the harness on request */
void rmxstub_clear_mex(void) { if (g_exit_fcn) g_exit_fcn(); }

static jmp_buf g_jmp;
static int g_armed = 0;
static char g_err[1024];
void mexErrMsgIdAndTxt(const char* id, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    const int k = snprintf(g_err, sizeof g_err, "%s: ", id);
    vsnprintf(g_err + k, sizeof g_err - (size_t)k, fmt, ap);
    va_end(ap);
    if (g_armed) longjmp(g_jmp, 1);
    fprintf(stderr, "%s\n", g_err);
    abort();
}

/* harness: run mexFunction, turning mexErrMsgIdAndTxt into a return code.  0 = ok, 1 = error (text via rmxstub_error) */
int rmxstub_call(int nlhs, mxArray** plhs, int nrhs, const mxArray** prhs) {
    g_err[0] = 0;
    for (int i = 0; i < nlhs; ++i) plhs[i] = NULL;
    if (setjmp(g_jmp)) {
        g_armed = 0;
        return 1;
    }
    g_armed = 1;
    mexFunction(nlhs, plhs, nrhs, prhs);
    g_armed = 0;
    return 0;
}
const char* rmxstub_error(void) { return g_err; }
mwSize rmxstub_ndim(const mxArray* a) { return a->ndim; }
mwSize rmxstub_dim(const mxArray* a, int i) { return a->dims[i]; }
int rmxstub_class(const mxArray* a) { return (int)a->cls; }
