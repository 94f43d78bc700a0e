// tune.h — launch-parameter cache (reference lib/tune.cpp:213-355, include/tune_quda.h): (volume, kernel, aux) -> launch parameters found
// by an in-process sweep, persisted as tunecache.tsv under QUDA_RESOURCE_PATH in the reference's text format.  See csrc/tune.cpp.
#pragma once

#include <string>

#include "qa_core.h"
#include "tune_key.h"

namespace quda {

struct TuneParam {   // reference include/tune_quda.h:20-60
  int block[3] = {0, 1, 1};
  int grid[3] = {0, 1, 1};
  int shared_bytes = 0;
  int aux[4] = {0, 0, 0, 0};   // stencil: aux.x y groups, aux.y link cache policy, aux.z store cache policy, aux.w reserved
  float time = 0;
  std::string comment;
};

// QudaInvertParam.tune == QUDA_TUNE_YES (setTuning) or QUDA_ENABLE_TUNING=1
bool tuningEnabled();
void loadTuneCache();    // once; initQuda
void saveTuneCache();    // rank 0, if entries were added; endQuda and after every sweep
const TuneParam *tuneLookup(const TuneKey &key);
void tuneStore(const TuneKey &key, const TuneParam &p);
int tuneCacheSize();
void tuneCountSweep();
long tuneSweeps();   // sweeps run by this process
void tuneCacheClear();   // forget everything, also the resource path (endQuda: the next initQuda reads the file again)

}  // namespace quda
