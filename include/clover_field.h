/* clover_field.h — reference header name (include/clover_field.h) for the field classes of this library, which live in fields.h */
#ifndef QUDA_AMD_FWD_CLOVER_FIELD_H
#define QUDA_AMD_FWD_CLOVER_FIELD_H
#include <fields.h>
#endif
