/*
 * qudaQKXTM_Kepler_utils.h — parameter structs of the QKXTM correlator drivers (reference include/qudaQKXTM_Kepler_utils.h:17-137),
 * as far as the entry points of qudaQKXTM_Kepler.h take them.  They are passed BY VALUE, so field order and array bounds are ABI;
 * enumerator values likewise.  One fixed layout: the deflation fields of qudaQKXTM_loopInfo that the reference compiles only
 * under HAVE_ARPACK are always present here (library and callers must agree, and there is one library build).
 */
#ifndef _QUDAQKXTM_KEPLER_UTILS_H
#define _QUDAQKXTM_KEPLER_UTILS_H

#include <quda.h>

#define QUDAQKXTM_DIM 4
#define MAX_NSOURCES 1000
#define MAX_NMOMENTA 5000
#define MAX_TSINK 10
#define MAX_DEFLSTEPS 10
#define MAX_PROJS 5

namespace quda {

  enum SOURCE_T { UNITY, RANDOM };
  enum CORR_SPACE { POSITION_SPACE, MOMENTUM_SPACE };
  enum FILE_WRITE_FORMAT { ASCII_FORM, HDF5_FORM };
  enum WHICHSPECTRUM { SR, LR, SM, LM, SI, LI };
  enum WHICHPARTICLE { PROTON, NEUTRON };

  typedef struct {
    int nsmearAPE;
    int nsmearGauss;
    double alphaAPE;
    double alphaGauss;
    int lL[QUDAQKXTM_DIM];
    int Nsources;
    int sourcePosition[MAX_NSOURCES][QUDAQKXTM_DIM];
    QudaPrecision Precision;
    int Q_sq;
    int Q_sq_loop;
    int Ntsink;
    int Nproj[MAX_TSINK];
    int traj;
    bool check_files;
    char *thrp_type[3];
    char *thrp_proj_type[5];
    char *baryon_type[10];
    char *meson_type[10];
    int tsinkSource[MAX_TSINK];
    int proj_list[MAX_TSINK][MAX_PROJS];
    int run3pt_src[MAX_NSOURCES];
    FILE_WRITE_FORMAT CorrFileFormat;
    SOURCE_T source_type;
    CORR_SPACE CorrSpace;
    bool HighMomForm;
    bool isEven;
    double kappa;
    double mu;
    double csw;
    double inv_tol;
  } qudaQKXTMinfo_Kepler;

  typedef struct {
    int PolyDeg;
    int nEv;
    int nKv;
    WHICHSPECTRUM spectrumPart;
    bool isACC;
    double tolArpack;
    int maxIterArpack;
    char arpack_logfile[512];
    double amin;
    double amax;
    bool isEven;
    bool isFullOp;
  } qudaQKXTM_arpackInfo;

  typedef struct {
    int Nstoch;
    unsigned long int seed;
    int Ndump;
    char loop_fname[512];
    int nSteps_defl;
    int deflStep[MAX_DEFLSTEPS];
    int traj;
    int Nprint;
    int Nmoms;
    int Qsq;
    FILE_WRITE_FORMAT FileFormat;
    char *loop_type[6];
    bool loop_oneD[6];
    bool useTSM;
    int TSM_NHP;
    int TSM_NLP;
    int TSM_NdumpHP;
    int TSM_NdumpLP;
    int TSM_NprintHP;
    int TSM_NprintLP;
    long int TSM_maxiter;
    double TSM_tol;
    bool HighMomForm;
    double kappa;
    double mu;
    double csw;
    double inv_tol;
  } qudaQKXTM_loopInfo;

}

#endif /* _QUDAQKXTM_KEPLER_UTILS_H */
