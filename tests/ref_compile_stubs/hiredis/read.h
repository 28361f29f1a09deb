/* Test-only stand-in for the un-vendored deps/hiredis submodule (see async.h). */
