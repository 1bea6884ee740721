//! qurious-hip: the MI355X (gfx950) execution backend of holicc/qurious' filter / hash-aggregate / hash-join path.
//!
//! ```ignore
//! use std::sync::Arc;
//! use qurious::dataframe::DataFrame;
//! use qurious_hip::HipQueryPlanner;
//!
//! // zero patches to the reference: DataFrame accepts any QueryPlanner (qurious/src/dataframe/mod.rs:21)
//! let planner = Arc::new(HipQueryPlanner::new()?);
//! let batches = DataFrame::new(logical_plan, planner).collect()?;
//! ```
//!
//! Layers: `ffi` (the `extern "C"` block of include/qhip.h), `lower` (LogicalExpr -> the flat `qhip_expr` arrays the C
//! ABI takes), `plan` (the `PhysicalPlan` nodes), `planner` (`HipQueryPlanner: QueryPlanner`).
pub mod ffi;
pub mod lower;
pub mod plan;
pub mod planner;

pub use plan::{DeviceTable, HipContext};
pub use planner::HipQueryPlanner;
