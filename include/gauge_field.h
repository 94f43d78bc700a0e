/* gauge_field.h — reference header name (include/gauge_field.h) for the field classes of this library, which live in fields.h */
#ifndef QUDA_AMD_FWD_GAUGE_FIELD_H
#define QUDA_AMD_FWD_GAUGE_FIELD_H
#include <fields.h>
#endif
