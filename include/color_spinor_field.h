/* color_spinor_field.h — reference header name (include/color_spinor_field.h) for the field classes of this library, which live in fields.h */
#ifndef QUDA_AMD_FWD_COLOR_SPINOR_FIELD_H
#define QUDA_AMD_FWD_COLOR_SPINOR_FIELD_H
#include <fields.h>
#endif
