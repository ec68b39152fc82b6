// QValue precision switch, as nlibs/tools/macro.h:5-6 of the reference (FSINGLE is the live setting there).
#ifndef SMF_MACRO_H_
#define SMF_MACRO_H_
typedef float QValue;
#define FSINGLE
#endif
