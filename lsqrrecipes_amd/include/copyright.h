// Header kept so that sources written against zivy/LSQRRecipes (#include "copyright.h") compile
// unchanged against this drop-in.  The MI355X implementation is original work; see README.md.
#ifndef _LSQR_AMD_COPYRIGHT_H_
#define _LSQR_AMD_COPYRIGHT_H_
#endif
