"""The Rust shim (rust/qurious-hip) is source only — the build image has no cargo / rustc — so nothing compiles
it here. What CAN be checked mechanically is checked: its FFI declarations against the C header they bind.

  * every function of `ffi.rs`'s `extern "C"` block is declared in include/qhip.h with the same number of arguments, the same
    argument names in the same order, pointer-ness and integer widths that agree, and the same return kind;
  * every `#[repr(C)]` struct has the header struct's fields, in order, with matching types;
  * the constant tables (status codes, type ids, expression / aggregate kinds) carry the header's values;
  * the operator / join-type codes lower.rs and plan.rs assign follow the reference enums' order (datatypes/operator.rs:4-20,
    common/join_type.rs:4-11), which is the order qhip.h declares;
  * the crate is complete: every module lib.rs names exists, and no file holds an elision marker.
"""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CRATE = os.path.join(ROOT, "rust", "qurious-hip")


def _strip_c_comments(text):
    return re.sub(r"/\*.*?\*/", " ", text, flags=re.S)


def _header():
    with open(os.path.join(ROOT, "include", "qhip.h")) as f:
        return _strip_c_comments(f.read())


def _rust(name):
    with open(os.path.join(CRATE, "src", name)) as f:
        text = f.read()
    return re.sub(r"//[^\n]*", "", text)


def _c_functions(h):
    """{name: (return type, [(type, name), ...])} of the header's prototypes"""
    out = {}
    for m in re.finditer(r"(?:^|[;}\n])\s*((?:const\s+)?[A-Za-z_][A-Za-z0-9_]*\s*\**)\s*(qhip_[a-z0-9_]+)\s*\(([^;{]*?)\)\s*;", h, flags=re.S):
        ret, name, args = m.group(1).strip(), m.group(2), m.group(3).strip()
        params = []
        if args and args != "void":
            for a in [x.strip() for x in args.split(",")]:
                a = re.sub(r"\s+", " ", a)
                pm = re.match(r"(.*?)([A-Za-z_][A-Za-z0-9_]*)$", a)
                params.append((pm.group(1).strip(), pm.group(2)))
        out[name] = (re.sub(r"\s+", " ", ret), params)
    return out


def _rust_functions(src):
    block = re.search(r'extern "C" \{(.*)\n\}', src, flags=re.S).group(1)
    out = {}
    for m in re.finditer(r"pub fn (qhip_[a-z0-9_]+)\s*\((.*?)\)\s*(?:->\s*([^;]+))?;", block, flags=re.S):
        name, args, ret = m.group(1), m.group(2), (m.group(3) or "").strip()
        params = []
        for a in [x.strip() for x in args.split(",") if x.strip()]:
            pname, ptype = [x.strip() for x in a.split(":", 1)]
            params.append((ptype, pname))
        out[name] = (ret, params)
    return out


# C type -> the Rust spellings that bind it
def _kind(c_type):
    t = re.sub(r"\bconst\b|\bstruct\b", "", c_type).replace(" ", "")
    stars = t.count("*")
    base = t.replace("*", "")
    return base, stars


RUST_OF_C = {
    "int": {"c_int", "i32"}, "int32_t": {"i32", "c_int"}, "int64_t": {"i64"}, "uint64_t": {"u64"}, "size_t": {"usize"}, "double": {"f64"},
    "char": {"c_char"}, "void": {"c_void"}, "uint32_t": {"u32"}, "uint8_t": {"u8"},
    "ArrowSchema": {"FFI_ArrowSchema"}, "ArrowArray": {"FFI_ArrowArray"},
}


def _matches(c_type, rust_type):
    base, stars = _kind(c_type)
    r = rust_type.replace(" ", "")
    r_stars = r.count("*const") + r.count("*mut")
    r_base = r.replace("*const", "").replace("*mut", "")
    if stars != r_stars:
        return False
    return r_base in RUST_OF_C.get(base, {base})     # qhip_* structs keep their C names


def test_every_rust_extern_matches_the_header():
    h, r = _c_functions(_header()), _rust_functions(_rust("ffi.rs"))
    assert len(r) >= 35
    for name, (ret, params) in r.items():
        assert name in h, f"{name} is not declared in include/qhip.h"
        c_ret, c_params = h[name]
        assert len(params) == len(c_params), f"{name}: {len(params)} arguments in ffi.rs, {len(c_params)} in qhip.h"
        for (rt, rn), (ct, cn) in zip(params, c_params):
            assert rn == cn or {rn, cn} <= {"input", "in"}, f"{name}: argument '{rn}' in ffi.rs is '{cn}' in qhip.h"
            assert _matches(ct, rt), f"{name}: argument {cn} is `{ct}` in qhip.h but `{rt}` in ffi.rs"
        if c_ret == "void":
            assert ret == "", name
        else:
            assert _matches(c_ret, ret), f"{name}: returns `{c_ret}` in qhip.h but `{ret}` in ffi.rs"
    # what a query host needs must be bound: context, tables, every operator, the exchange
    for must in ("qhip_ctx_create", "qhip_table_from_arrow", "qhip_table_to_arrow", "qhip_filter_execute", "qhip_hash_aggregate_execute",
                 "qhip_hash_join_execute", "qhip_nested_loop_join_execute", "qhip_cross_join_execute", "qhip_projection_execute",
                 "qhip_sort_execute", "qhip_limit_execute", "qhip_ctx_allow_deferred_sizes", "qhip_partition_by_key", "qhip_comm_create",
                 "qhip_exchange_tables", "qhip_all_gather_table"):
        assert must in r, must


def _c_structs(h):
    out = {}
    for m in re.finditer(r"typedef struct (qhip_[a-z_]+) \{(.*?)\} \1;", h, flags=re.S):
        fields = []
        for decl in [d.strip() for d in m.group(2).split(";") if d.strip()]:
            decl = re.sub(r"\s+", " ", decl)
            tm = re.match(r"(.*?)([A-Za-z_][A-Za-z0-9_]*(?:\[\d+\])?(?:\s*,\s*[A-Za-z_][A-Za-z0-9_]*)*)$", decl)
            ctype, names = tm.group(1).strip(), [n.strip() for n in tm.group(2).split(",")]
            for n in names:
                fields.append((ctype, n))
        out[m.group(1)] = fields
    return out


def _rust_structs(src):
    out = {}
    for m in re.finditer(r"#\[repr\(C\)\]\s*(?:#\[derive\([^\]]*\)\]\s*)?pub struct (qhip_[a-z_]+) \{(.*?)\n\}", src, flags=re.S):
        fields = []
        for line in [x.strip() for x in m.group(2).split(",") if x.strip()]:
            fm = re.match(r"(?:pub\s+)?([A-Za-z_][A-Za-z0-9_]*)\s*:\s*(.+)$", line, flags=re.S)
            fields.append((fm.group(2).strip(), fm.group(1)))
        out[m.group(1)] = fields
    return out


def test_repr_c_structs_have_the_headers_fields_in_order():
    h, r = _c_structs(_header()), _rust_structs(_rust("ffi.rs"))
    for name in ("qhip_dtype", "qhip_expr", "qhip_agg", "qhip_exec_stats", "qhip_comm_stats"):
        assert name in r and name in h, name
        c_fields, r_fields = h[name], r[name]
        assert [n.split("[")[0] for _, n in c_fields] == [n for _, n in r_fields], f"{name}: field order differs"
        for (ct, cn), (rt, rn) in zip(c_fields, r_fields):
            am = re.match(r"(.+)\[(\d+)\]$", cn)
            if am:   # char main_kernel_name[64] <-> [c_char; 64]
                assert rt.replace(" ", "") == f"[c_char;{am.group(2)}]", (name, cn, rt)
            else:
                assert _matches(ct, rt), f"{name}.{cn}: `{ct}` in qhip.h but `{rt}` in ffi.rs"
    # the opaque handles are opaque on both sides
    for handle in ("qhip_ctx", "qhip_table", "qhip_comm"):
        assert re.search(rf"typedef struct {handle} {handle};", _header()) and re.search(rf"pub struct {handle} \{{\s*_private: \[u8; 0\]", _rust("ffi.rs"))


def test_constants_carry_the_headers_values():
    h = _header()
    src = _rust("ffi.rs")
    consts = {m.group(1): int(m.group(2)) for m in re.finditer(r"pub const (QHIP_[A-Z0-9_]+): [a-z_0-9]+ = (\d+);", src)}
    enums = {}
    for m in re.finditer(r"typedef enum [a-z_]+ \{(.*?)\}", h, flags=re.S):
        nxt = 0
        for item in [x.strip() for x in m.group(1).split(",") if x.strip()]:
            if "=" in item:
                k, v = [x.strip() for x in item.split("=")]
                nxt = int(v)
            else:
                k = item
            enums[k] = nxt
            nxt += 1
    assert len(consts) >= 38
    for k, v in consts.items():
        assert enums.get(k) == v, f"{k} = {v} in ffi.rs, {enums.get(k)} in qhip.h"
    # operator and join-type codes follow the header's enum order (which is the reference enums' order)
    ops = re.search(r"pub fn operator_code.*?\{(.*?)\n\}", _rust("lower.rs"), flags=re.S).group(1)
    got = {m.group(1).upper(): int(m.group(2)) for m in re.finditer(r"Operator::(\w+) => (\d+)", ops)}
    assert got == {k[len("QHIP_OP_"):]: v for k, v in enums.items() if k.startswith("QHIP_OP_")}
    jt = re.search(r"pub fn join_type_code.*?\{(.*?)\n\}", _rust("plan.rs"), flags=re.S).group(1)
    got = {re.sub(r"(?<!^)(?=[A-Z])", "_", m.group(1)).upper(): int(m.group(2)) for m in re.finditer(r"JoinType::(\w+) => (\d+)", jt)}
    assert got == {k[len("QHIP_JOIN_"):]: v for k, v in enums.items() if k.startswith("QHIP_JOIN_")}


def test_the_crate_is_complete():
    for f in ("Cargo.toml", "build.rs", "src/lib.rs", "src/ffi.rs", "src/lower.rs", "src/plan.rs", "src/planner.rs"):
        path = os.path.join(CRATE, f)
        assert os.path.exists(path), f
        text = open(path).read()
        for marker in ("todo!(", "unimplemented!(", "analogous", "elided", "// ..."):
            assert marker not in text, (f, marker)
    lib = open(os.path.join(CRATE, "src", "lib.rs")).read()
    assert set(re.findall(r"pub mod (\w+);", lib)) == {"ffi", "lower", "plan", "planner"}
    planner = _rust("planner.rs")
    assert "impl QueryPlanner for HipQueryPlanner" in planner and "fn create_physical_plan" in planner and "fn create_physical_expr" in planner
    # one HIP node per operator of the path, each implementing the reference's trait
    plan = _rust("plan.rs")
    for node in ("HipScan", "HipFilter", "HipAggregate", "HipHashJoin", "HipNestedLoopJoin", "HipProjection", "HipSort", "HipLimit"):
        assert f"impl PhysicalPlan for {node}" in plan and f"impl HipNode for {node}" in plan, node
    # every extern the nodes call is declared in ffi.rs
    declared = set(_rust_functions(_rust("ffi.rs")))
    for called in set(re.findall(r"\b(qhip_[a-z0-9_]+)\s*\(", plan)):
        assert called in declared, called


def test_the_table_cache_is_keyed_on_the_identity_of_the_data():
    """`(provider address, batches, rows)` is the same key after DELETE-all + INSERT of the same shape
    (datasource/memory.rs:104-137) or for a new provider at a freed address — the HIP path would answer from the old table.
    The entry must be validated against the provider's liveness and the very arrays it was uploaded from."""
    src = _rust("plan.rs")
    fn = src[src.index("pub fn table_of"):]
    fn = fn[:fn.index("\n    }\n") + 7]
    # the old key is gone ...
    assert "batches.len(), rows)" not in src and "HashMap<(usize, usize, usize)" not in src
    # ... the entry remembers a Weak to the provider and the batches it was made from, and both are checked before reuse
    entry = src[src.index("struct CachedTable"):src.index("fn same_batches")]
    assert "Weak<dyn TableProvider>" in entry and "Vec<RecordBatch>" in entry and "Arc<DeviceTable>" in entry
    assert "upgrade()" in fn and "Arc::ptr_eq(&p, source)" in fn and "same_batches(&e.batches, &batches)" in fn
    assert "Arc::downgrade(source)" in fn and "strong_count() > 0" in fn
    same = src[src.index("fn same_batches"):src.index("unsafe impl Send for HipContext")]
    assert "Arc::ptr_eq(p, q)" in same and "num_rows()" in same and "a.len() == b.len()" in same
    # the rule is written down where a maintainer looks for it
    integ = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    assert "memory.rs:104-137" in integ and "Arc::ptr_eq" in integ and "Weak<dyn TableProvider>" in integ


def test_the_python_mirror_applies_the_same_invalidation_rule():
    """MemoryTable.device_table re-uploads when the batch list no longer holds the objects the copy was made from (checked
    without a GPU: DeviceTable.from_batches is replaced by a recorder)"""
    import pyarrow as pa
    from qurious_amd import plan as P
    made = []
    real = P.DeviceTable.from_batches
    P.DeviceTable.from_batches = staticmethod(lambda ctx, schema, data, lazy=False: made.append(list(data)) or object())
    real_ctx = P.get_context
    P.get_context = lambda: None
    try:
        schema = pa.schema([pa.field("v", pa.int64())])
        mk = lambda vals: pa.RecordBatch.from_arrays([pa.array(vals, type=pa.int64())], schema=schema)   # noqa: E731
        t = P.MemoryTable.try_new(schema, [mk([1, 2, 3])])
        a = t.device_table()
        assert t.device_table() is a and len(made) == 1                 # unchanged data: one upload
        t.delete()                                                      # DELETE everything ...
        t.insert([mk([7, 8, 9])])                                       # ... INSERT the same shape: same table, 1 batch, 3 rows
        b = t.device_table()
        assert b is not a and len(made) == 2 and made[1][0].column(0).to_pylist() == [7, 8, 9]
        t.insert([mk([4])])
        assert t.device_table() is not b and len(made) == 3             # an append is new data too
    finally:
        P.DeviceTable.from_batches = real
        P.get_context = real_ctx
