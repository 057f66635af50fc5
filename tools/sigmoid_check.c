// Accuracy of the two sigmoid formulations of csrc/flow_common.h against quad precision (gcc tools/sigmoid_check.c -lquadmath -lm):
// the degree-11 polynomial of exp(r) and the rational form folded into the division, hi/lo and single-constant range reduction.
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <quadmath.h>
static double rcp_ref(double t){ /* emulate v_rcp_f64 with 4.6e-8 error + third-order step */
    double y = (double)(float)(1.0/t); /* ~6e-8 */
    double u = fma(-t,y,1.0); return fma(fma(u,u,u),y,y); }
static double sig_cur(double z){
    double a = fmin(-z,700.0);
    double n = rint(a*1.4426950408889634074);
    double r = fma(-n,6.93147180369123816490e-01,a); r = fma(-n,1.90821492927058770002e-10,r);
    double p = 2.5100375832561234e-08;
    const double C[]={2.7620075879983367e-07,2.7557268480310024e-06,2.4801521322368692e-05,0.00019841269863040545,0.0013888888917196719,0.008333333333330065,0.041666666666624164,0.16666666666666669,0.5000000000000001,1.0,1.0};
    for(int i=0;i<11;i++) p=fma(p,r,C[i]);
    double t = 1.0+ldexp(p,(int)n);
    return rcp_ref(t);
}
static double sig_new(double z, int lo){
    double a = fmin(-z,700.0);
    double n = rint(a*1.4426950408889634074);
    double r;
    if (lo) { r = fma(-n,6.93147180369123816490e-01,a); r = fma(-n,1.90821492927058770002e-10,r); }
    else r = fma(-n,6.931471805599453094e-01,a);
    double s = r*r;
    double P = 4.13813679705723846039e-08;
    P = fma(P,s,-1.65339022054652515390e-06); P = fma(P,s,6.61375632143793436117e-05); P = fma(P,s,-2.77777777770155933842e-03); P = fma(P,s,1.66666666666666019037e-01);
    double c = fma(-s,P,r);
    double m = 2.0 - c;
    double N = fma(2.0,r,m);
    double E = ldexp(N,(int)n);
    double den = m + E;
    double y = rcp_ref(den);
    return m*y;
}
// round-4 form: n by the 1.5 * 2^52 trick (the FMA rounds a log2 e to an integer), r = a - n ln2 in one FMA (lo = 0) or hi / lo
static double sig_magic(double z, int lo){
    double a = fmin(-z,700.0);
    double t = fma(a,1.4426950408889634074,6755399441055744.0);
    double n = t - 6755399441055744.0;
    double r;
    if (lo) { r = fma(-n,6.93147180369123816490e-01,a); r = fma(-n,1.90821492927058770002e-10,r); }
    else r = fma(-n,6.93147180559945309417e-01,a);
    double s = r*r;
    double P = 4.13813679705723846039e-08;
    P = fma(P,s,-1.65339022054652515390e-06); P = fma(P,s,6.61375632143793436117e-05); P = fma(P,s,-2.77777777770155933842e-03); P = fma(P,s,1.66666666666666019037e-01);
    double c = fma(-s,P,r);
    double m = 2.0 - c;
    double N = fma(2.0,r,m);
    double E = ldexp(N,(int)n);
    double den = m + E;
    return m*rcp_ref(den);
}
// absolute errors of sigma, h = z sigma and act' = sigma + h (1 - sigma) in units of 2^-53 (the rounding error of a value near 1)
static void abs_check(void){
    srand48(11); double w[3][3]={{0}};
    for(long i=0;i<4000000;i++){
        double z = (drand48()*2-1)* (i%4==0?40.0: i%4==1?8.0: 3.0);
        __float128 sg = 1.0Q/(1.0Q+expq(-(__float128)z)), hq = (__float128)z*sg, dq = sg + hq*(1.0Q-sg);
        double v[3]={sig_new(z,1),sig_magic(z,1),sig_magic(z,0)};
        for(int k=0;k<3;k++){
            double h = z*v[k], d = fma(h,1.0-v[k],v[k]);
            double e0 = fabs((double)((__float128)v[k]-sg))*0x1p53, e1 = fabs((double)((__float128)h-hq))*0x1p53, e2 = fabs((double)((__float128)d-dq))*0x1p53;
            if(e0>w[k][0])w[k][0]=e0; if(e1>w[k][1])w[k][1]=e1; if(e2>w[k][2])w[k][2]=e2;
        }
    }
    const char* nm[3]={"round 3 (rint, hi/lo)","magic rounding, hi/lo","magic rounding, one-FMA ln2"};
    for(int k=0;k<3;k++) printf("%-30s max |err| / 2^-53: sigma %.3f  h %.3f  act' %.3f\n",nm[k],w[k][0],w[k][1],w[k][2]);
}
int main(){
    abs_check();
    srand48(7); double w[3]={0,0,0}; double ws[3]={0,0,0};
    for(long i=0;i<4000000;i++){
        double z = (drand48()*2-1)* (i%4==0?40.0: i%4==1?8.0: 3.0);
        __float128 ref = 1.0Q/(1.0Q+expq(-(__float128)z));
        double rd = (double)ref; double ulp = nextafter(rd,INFINITY)-rd;
        double v[3]={sig_cur(z),sig_new(z,1),sig_new(z,0)};
        for(int k=0;k<3;k++){ double e = fabs((double)((__float128)v[k]-ref))/ulp; if(e>w[k])w[k]=e; ws[k]+=e; }
    }
    printf("max ulp: cur %.3f new(hi/lo) %.3f new(single ln2) %.3f; mean %.3f %.3f %.3f\n",w[0],w[1],w[2],ws[0]/4e6,ws[1]/4e6,ws[2]/4e6);
    return 0;
}
