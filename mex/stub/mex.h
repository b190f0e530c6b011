/*
 * mex.h -- stand-in for MATLAB's <mex.h> / <matrix.h>, declaring exactly the part of the C Matrix API and MEX API that
 * mex/redmax_hip_mex.c uses, with MATLAB's documented signatures (R2018a+).  It exists because MATLAB is not available
 * where this repository is built and tested: tests/test_mex_gateway.py compiles the gateway against this header, links it
 * with mex_stub.c (a small host-memory implementation of these functions) and libredmax_hip.so, and drives mexFunction
 * through ctypes.  Inside MATLAB none of this is used: `mex` supplies the real header and library.
 */
#ifndef RMX_STUB_MEX_H
#define RMX_STUB_MEX_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct mxArray_tag mxArray;
typedef size_t mwSize;
typedef size_t mwIndex;
typedef enum { mxREAL = 0, mxCOMPLEX = 1 } mxComplexity;
typedef enum {
    mxUNKNOWN_CLASS = 0, mxCELL_CLASS, mxSTRUCT_CLASS, mxLOGICAL_CLASS, mxCHAR_CLASS, mxVOID_CLASS, mxDOUBLE_CLASS,
    mxSINGLE_CLASS, mxINT8_CLASS, mxUINT8_CLASS, mxINT16_CLASS, mxUINT16_CLASS, mxINT32_CLASS, mxUINT32_CLASS,
    mxINT64_CLASS, mxUINT64_CLASS
} mxClassID;

/* creation */
mxArray* mxCreateDoubleMatrix(mwSize m, mwSize n, mxComplexity flag);
mxArray* mxCreateDoubleScalar(double value);
mxArray* mxCreateNumericMatrix(mwSize m, mwSize n, mxClassID classid, mxComplexity flag);
mxArray* mxCreateNumericArray(mwSize ndim, const mwSize* dims, mxClassID classid, mxComplexity flag);
mxArray* mxCreateStructMatrix(mwSize m, mwSize n, int nfields, const char** fieldnames);
mxArray* mxCreateString(const char* str);
void mxDestroyArray(mxArray* pa);
/* access */
double* mxGetPr(const mxArray* pa);
void* mxGetData(const mxArray* pa);
double mxGetScalar(const mxArray* pa);
size_t mxGetNumberOfElements(const mxArray* pa);
size_t mxGetM(const mxArray* pa);
size_t mxGetN(const mxArray* pa);
int mxGetString(const mxArray* pa, char* buf, mwSize buflen);
mxArray* mxGetField(const mxArray* pa, mwIndex index, const char* fieldname);
void mxSetField(mxArray* pa, mwIndex index, const char* fieldname, mxArray* value);
/* predicates */
int mxIsStruct(const mxArray* pa);
int mxIsDouble(const mxArray* pa);
int mxIsInt32(const mxArray* pa);
int mxIsUint64(const mxArray* pa);
int mxIsComplex(const mxArray* pa);
int mxIsEmpty(const mxArray* pa);
/* memory */
void* mxCalloc(size_t n, size_t size);
void mxFree(void* ptr);
void mexMakeMemoryPersistent(void* ptr);
int mexAtExit(void (*exit_fcn)(void));
/* errors: does not return (longjmps back into rmxstub_call) */
void mexErrMsgIdAndTxt(const char* id, const char* fmt, ...);

void mexFunction(int nlhs, mxArray* plhs[], int nrhs, const mxArray* prhs[]);

#ifdef __cplusplus
}
#endif
#endif
