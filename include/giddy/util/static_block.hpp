// static_block { ... } — runs a block once during static initialisation of the translation unit.
// Same spelling as the reference's src/util/static_block.h:69-77 so wrapper sources read alike;
// implemented here with a uniquely named file-scope object whose constructor runs the body.
#pragma once

#define GIDDY_SB_CAT_(a, b) a##b
#define GIDDY_SB_CAT(a, b) GIDDY_SB_CAT_(a, b)
#define GIDDY_SB_IMPL(fn, obj)                                \
	static void fn();                                         \
	namespace { struct GIDDY_SB_CAT(obj, _t) { GIDDY_SB_CAT(obj, _t)() { fn(); } } obj; } \
	static void fn()
#define static_block GIDDY_SB_IMPL(GIDDY_SB_CAT(giddy_static_block_fn_, __COUNTER__), GIDDY_SB_CAT(giddy_static_block_obj_, __COUNTER__))
