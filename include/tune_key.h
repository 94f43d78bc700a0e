/*
 * tune_key.h — identification of the most recently launched kernel, quoted by the error messages of util_quda.h.
 * Counterpart of the reference's include/tune_key.h (struct layout is ABI: getLastTuneKey() returns it by value, and objects
 * built against the reference header call it from errorQuda).  This library has no autotuner; the key names the last stencil
 * launch geometry (csrc/dslash.hip) instead of a tune-cache entry.
 */
#ifndef _TUNE_KEY_H
#define _TUNE_KEY_H

#include <cstring>

namespace quda {
  struct TuneKey {
    static const int volume_n = 32;
    static const int name_n = 384;
    static const int aux_n = 256;
    char volume[volume_n];
    char name[name_n];
    char aux[aux_n];
    TuneKey() { volume[0] = name[0] = aux[0] = 0; }
    TuneKey(const char v[], const char n[], const char a[] = "type=default") {
      strncpy(volume, v, volume_n - 1); volume[volume_n - 1] = 0;
      strncpy(name, n, name_n - 1); name[name_n - 1] = 0;
      strncpy(aux, a, aux_n - 1); aux[aux_n - 1] = 0;
    }
    bool operator<(const TuneKey &o) const {
      int c = std::strcmp(volume, o.volume);
      if (c) return c < 0;
      c = std::strcmp(name, o.name);
      if (c) return c < 0;
      return std::strcmp(aux, o.aux) < 0;
    }
  };
}

quda::TuneKey getLastTuneKey();

#endif /* _TUNE_KEY_H */
