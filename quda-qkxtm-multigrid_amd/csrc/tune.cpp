// tune.cpp — launch-parameter cache of the stencil (reference lib/tune.cpp:213-355: tunecache.tsv under QUDA_RESOURCE_PATH, read by
// loadTuneCache at start-up, written by saveTuneCache when entries were added; TuneKey = (volume, kernel name, aux), TuneParam = block /
// grid / shared bytes / aux int4 / time / comment).
//
// What is cached here.  The launch geometry of the stencil is a closed-form function of the lattice (dslash.hip launchDslash), but four
// knobs have a best value that flips with volume, precision and partitioning — measured in rounds 2 and 3: the cache policy of the
// 16-bit link loads (+13 % at 32^4, -6 % at 48^3 x 96), non-temporal output stores (+5 % from 2^18 sites up, -5 % on the 8-GPU
// sub-lattice), the y groups of the plane-tiled order (fp64 at 48^3 x 96) and the block size.  Heuristics stand in for them; with tuning
// enabled (QudaInvertParam.tune = QUDA_TUNE_YES, or QUDA_ENABLE_TUNING=1 as in the reference) the first launch of a key instead times
// the candidates INTERLEAVED — A B C ... A B C ..., three rounds, the minimum per candidate, because consecutive runs of one candidate
// share whatever placement / clock state the device is in (tools/policy_interleaved.sh, profiles/r03_link_policy_interleaved_one_box.log)
// — keeps the fastest, and the table is persisted in the reference's text format: the knobs travel in block.x and aux.x .. aux.w.
#include "tune.h"

#include <sys/stat.h>
#include <fcntl.h>
#include <unistd.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <fstream>
#include <iomanip>
#include <map>
#include <sstream>
#include <string>

QudaTune getTuning();   // util_quda.h (global namespace, as in the reference): set from QudaInvertParam.tune by the API entry points

namespace quda {

static const char *kTuneVersion = "0.8.0-amd";        // first header token after "tunecache" (the reference writes its version there)
static const char *kTuneHash = "gfx950-dslash-r4";    // build identification: a cache of another kernel generation is discarded, not trusted

static std::map<TuneKey, TuneParam> g_cache;
static std::string g_resourcePath;
static size_t g_initialSize = 0;
static bool g_loaded = false;

bool tuningEnabled() {
  static int env = -1;
  if (env < 0) { const char *e = getenv("QUDA_ENABLE_TUNING"); env = e ? atoi(e) : 0; }
  return env != 0 || ::getTuning() == QUDA_TUNE_YES;
}

// reference deserializeTuneCache, lib/tune.cpp:124-160: one entry per line, tab separated
static void deserialize(std::istream &in) {
  std::string line;
  while (in.good()) {
    getline(in, line);
    if (line.empty()) continue;
    std::stringstream ls(line);
    std::string v, n, a;
    TuneParam p;
    getline(ls, v, '\t'); getline(ls, n, '\t'); getline(ls, a, '\t');
    // the reference right-aligns the volume column: strip the padding
    const size_t b = v.find_first_not_of(' ');
    if (b == std::string::npos) continue;
    v = v.substr(b);
    ls >> p.block[0] >> p.block[1] >> p.block[2] >> p.grid[0] >> p.grid[1] >> p.grid[2] >> p.shared_bytes >> p.aux[0] >> p.aux[1] >> p.aux[2] >> p.aux[3] >> p.time;
    if (ls.fail()) errorQuda("Bad format in tunecache entry '%s'", line.c_str());
    ls.ignore(1);
    getline(ls, p.comment);
    g_cache[TuneKey(v.c_str(), n.c_str(), a.c_str())] = p;
  }
}
// reference serializeTuneCache, lib/tune.cpp:103-121
static void serialize(std::ostream &out) {
  for (const auto &e : g_cache) {
    const TuneKey &k = e.first;
    const TuneParam &p = e.second;
    out << std::setw(16) << k.volume << "\t" << k.name << "\t" << k.aux << "\t";
    out << p.block[0] << "\t" << p.block[1] << "\t" << p.block[2] << "\t" << p.grid[0] << "\t" << p.grid[1] << "\t" << p.grid[2] << "\t";
    out << p.shared_bytes << "\t" << p.aux[0] << "\t" << p.aux[1] << "\t" << p.aux[2] << "\t" << p.aux[3] << "\t" << p.time << "\t" << p.comment << std::endl;
  }
}

void loadTuneCache() {
  if (g_loaded) return;
  g_loaded = true;
  const char *path = getenv("QUDA_RESOURCE_PATH");
  struct stat pstat;
  if (!path) {
    if (tuningEnabled()) {
      warningQuda("Environment variable QUDA_RESOURCE_PATH is not set.");
      warningQuda("Caching of tuned parameters will be disabled.");
    }
    return;
  }
  if (stat(path, &pstat) || !S_ISDIR(pstat.st_mode)) {
    warningQuda("The path \"%s\" specified by QUDA_RESOURCE_PATH does not exist or is not a directory.", path);
    warningQuda("Caching of tuned parameters will be disabled.");
    return;
  }
  g_resourcePath = path;
  const std::string cachePath = g_resourcePath + "/tunecache.tsv";
  std::ifstream f(cachePath.c_str());
  if (!f) {
    if (tuningEnabled()) warningQuda("Cache file not found.  All kernels will be re-tuned (if tuning is enabled).");
    return;
  }
  std::string line, token;
  getline(f, line);
  std::stringstream ls(line);
  ls >> token;
  if (token.compare("tunecache")) errorQuda("Bad format in %s", cachePath.c_str());
  std::string version, gitversion, hash;
  ls >> version >> gitversion >> hash;
  if (version.compare(kTuneVersion) || hash.compare(kTuneHash)) {
    // the reference aborts here ("Please delete this file"); a stale table is only a slower start, so it is ignored and overwritten
    warningQuda("Cache file %s was written by another build (%s %s): ignored, it will be replaced", cachePath.c_str(), version.c_str(), hash.c_str());
    return;
  }
  getline(f, line);   // the blank line behind the time stamp
  getline(f, line);   // the description line
  deserialize(f);
  g_initialSize = g_cache.size();
  if (getVerbosity() >= QUDA_SUMMARIZE) printfQuda("Loaded %d sets of cached parameters from %s\n", (int)g_initialSize, cachePath.c_str());
}

void saveTuneCache() {
  if (g_resourcePath.empty() || commGrid().rank != 0) return;
  if (g_cache.size() == g_initialSize) return;
  const std::string lockPath = g_resourcePath + "/tunecache.lock";
  const int lock = open(lockPath.c_str(), O_WRONLY | O_CREAT | O_EXCL, 0666);
  if (lock == -1) {
    warningQuda("Unable to lock cache file.  Tuned launch parameters will not be cached to disk.  If you are certain that no other instances of QUDA are accessing this filesystem, please manually remove %s", lockPath.c_str());
    return;
  }
  const char msg[] = "If no instances of applications using QUDA are running,\nthis lock file shouldn't be here and is safe to delete.";
  if (write(lock, msg, sizeof(msg)) == -1) warningQuda("Unable to write to lock file for some bizarre reason");
  const std::string cachePath = g_resourcePath + "/tunecache.tsv";
  std::ofstream f(cachePath.c_str());
  if (getVerbosity() >= QUDA_SUMMARIZE) printfQuda("Saving %d sets of cached parameters to %s\n", (int)g_cache.size(), cachePath.c_str());
  time_t now;
  time(&now);
  f << "tunecache\t" << kTuneVersion << "\t" << kTuneVersion << "\t" << kTuneHash << "\t# Last updated " << ctime(&now) << std::endl;
  f << std::setw(16) << "volume" << "\tname\taux\tblock.x\tblock.y\tblock.z\tgrid.x\tgrid.y\tgrid.z\tshared_bytes\taux.x\taux.y\taux.z\taux.w\ttime\tcomment" << std::endl;
  serialize(f);
  f.close();
  close(lock);
  remove(lockPath.c_str());
  g_initialSize = g_cache.size();
}

const TuneParam *tuneLookup(const TuneKey &key) {
  loadTuneCache();
  const auto it = g_cache.find(key);
  return it == g_cache.end() ? nullptr : &it->second;
}
void tuneStore(const TuneKey &key, const TuneParam &p) {
  loadTuneCache();
  g_cache[key] = p;
}
int tuneCacheSize() { loadTuneCache(); return (int)g_cache.size(); }
static long g_sweeps = 0;
void tuneCountSweep() { g_sweeps++; }
long tuneSweeps() { return g_sweeps; }
void tuneCacheClear() { g_cache.clear(); g_initialSize = 0; g_loaded = false; g_resourcePath.clear(); }

}  // namespace quda

using namespace quda;

// C access for callers and tests (quda_amd_ext.h): the table and its file, without a device
extern "C" {
int qudaAmdTuneCacheLoad(void) { tuneCacheClear(); loadTuneCache(); return tuneCacheSize(); }
void qudaAmdTuneCacheSave(void) { saveTuneCache(); }
void qudaAmdTuneCacheStore(const char *volume, const char *name, const char *aux, const int param[11], float time, const char *comment) {
  TuneParam p;
  for (int i = 0; i < 3; i++) { p.block[i] = param[i]; p.grid[i] = param[3 + i]; }
  p.shared_bytes = param[6];
  for (int i = 0; i < 4; i++) p.aux[i] = param[7 + i];
  p.time = time; p.comment = comment ? comment : "";
  tuneStore(TuneKey(volume, name, aux), p);
}
int qudaAmdTuneCacheLookup(const char *volume, const char *name, const char *aux, int param[11], float *time) {
  const TuneParam *p = tuneLookup(TuneKey(volume, name, aux));
  if (!p) return 0;
  for (int i = 0; i < 3; i++) { param[i] = p->block[i]; param[3 + i] = p->grid[i]; }
  param[6] = p->shared_bytes;
  for (int i = 0; i < 4; i++) param[7 + i] = p->aux[i];
  if (time) *time = p->time;
  return 1;
}
long qudaAmdTuneSweeps(void) { return tuneSweeps(); }
}
