/* Host build of include/lupin_detmath.h for the tests (compiled on the fly, -ffp-contract=off). */
#include "../include/lupin_detmath.h"
void detmath_eval(int fn, unsigned n, const float *x, const float *y, float *out)
{
    for (unsigned i = 0; i < n; i++)
    {
        float a = x[i], b = y[i], r;
        switch (fn)
        {
        case 0: r = lpm_sinf(a); break;
        case 1: r = lpm_cosf(a); break;
        case 2: r = lpm_atanf(a); break;
        case 3: r = lpm_atan2f(a, b); break;
        case 4: r = lpm_acosf(a); break;
        case 5: r = lpm_expf(a); break;
        case 6: r = lpm_logf(a); break;
        case 7: r = lpm_powf(a, b); break;
        case 8: r = a / b; break;
        case 9: r = sqrtf(a); break;
        default: r = 0.0f; break;
        }
        out[i] = r;
    }
}
