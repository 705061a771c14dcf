// oracle/shim/profi_decls.h — the reference includes the un-vendored "profi" profiler
// (stdafx.h:24-25).  Empty stand-in; test infrastructure only.
#pragma once
