// TEST INFRASTRUCTURE ONLY. Not part of the product.
//
// The two LAPACK routines the reference's uncertainty.c calls (uncertainty.c:788-792,
// 1501, 1519): packed Cholesky factorization and solve, lower storage, on its 6x6
// Jcross_t Jcross. LAPACK is not installed in this image; these follow LAPACK's documented
// semantics for uplo='L' (column-major packed lower triangle: AP[i + j(2n-j-1)/2], i>=j)
// so that the reference's own _mrcal_drt_cross_reprojection__dbpacked() runs, unmodified,
// as the checker of mrcal_amd's. tests/test_oracle_lapack_stub.py pins them against numpy.
//
// Note the reference's calling convention: it accumulates the UPPER triangle in row-major
// packed order and hands it over as 'L' column-major packed - the same bytes.
#include <math.h>

int dpptrf_(char* uplo, int* n_, double* ap, int* info)
{
    const int n = *n_;
    *info = 0;
    if(!(uplo[0] == 'L' || uplo[0] == 'l')) { *info = -1; return 0; }
#define AP(i,j) ap[(i) + (j)*(2*n-(j)-1)/2]
    for(int j=0; j<n; j++)
    {
        double d = AP(j,j);
        for(int k=0; k<j; k++) d -= AP(j,k)*AP(j,k);
        if(!(d > 0.0)) { *info = j+1; return 0; }
        d = sqrt(d);
        AP(j,j) = d;
        for(int i=j+1; i<n; i++)
        {
            double s = AP(i,j);
            for(int k=0; k<j; k++) s -= AP(i,k)*AP(j,k);
            AP(i,j) = s/d;
        }
    }
    return 0;
}

int dpptrs_(char* uplo, int* n_, int* nrhs_, double* ap, double* b, int* ldb_, int* info)
{
    const int n = *n_, nrhs = *nrhs_, ldb = *ldb_;
    *info = 0;
    if(!(uplo[0] == 'L' || uplo[0] == 'l')) { *info = -1; return 0; }
    for(int r=0; r<nrhs; r++)
    {
        double* x = &b[r*ldb];
        for(int i=0; i<n; i++)      // L y = b
        {
            double s = x[i];
            for(int k=0; k<i; k++) s -= AP(i,k)*x[k];
            x[i] = s/AP(i,i);
        }
        for(int i=n-1; i>=0; i--)   // Lt x = y
        {
            double s = x[i];
            for(int k=i+1; k<n; k++) s -= AP(k,i)*x[k];
            x[i] = s/AP(i,i);
        }
    }
#undef AP
    return 0;
}
