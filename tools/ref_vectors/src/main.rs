//! ref_vectors -- runs the reference's own `integrate()` (gravitas-core, geodesic/mod.rs:180-253) on
//! the inputs of this repo's golden ray sets and writes end states, step counts, termination classes,
//! Hamiltonian drifts and (for the first ray of every case) the recorded path as JSON.
//!
//!   python tools/ref_vectors/export_inputs.py            # tests/golden/rays_v1.npz -> inputs.txt
//!   cargo run --release -- inputs.txt ../../tests/golden/ref_rays_v1.json
//!
//! Every f64 travels as the 16 hex digits of its bit pattern, so nothing is lost in either direction.
//! No dependencies besides gravitas-core itself.
use std::env;
use std::fmt::Write as _;
use std::fs;

use gravitas::geodesic::{integrate, GeodesicState, IntegrationMethod, IntegrationOptions, Trajectory};
use gravitas::metric::{Kerr, Metric, Schwarzschild};

fn f64_from_hex(s: &str) -> f64 {
    f64::from_bits(u64::from_str_radix(s, 16).expect("16 hex digits"))
}

fn hex(x: f64) -> String {
    format!("\"{:016x}\"", x.to_bits())
}

struct Case {
    key: String,
    kind: u32,
    spin: f64,
    method: u32,
    tolerance: f64,
    max_steps: usize,
    step_size: f64,
    escape_radius: f64,
    renormalize_interval: usize,
    initial_step: f64,
    rays: Vec<GeodesicState>,
}

fn field_of<'a>(w: &[&'a str], name: &str) -> &'a str {
    let i = w.iter().position(|x| *x == name).unwrap_or_else(|| panic!("missing {}", name));
    w[i + 1]
}

/// inputs.txt: a header line per case,
///   case <key> kind <0 BL | 1 KS | 2 Schwarzschild> spin <hex> method <0 rkf45 | 1 rk4 | 2 symplectic>
///        tol <hex> max_steps <int> step <hex> esc <hex> renorm <int> h0 <hex> n <int>
/// followed by n lines of 8 hex words (t r theta phi p_t p_r p_theta p_phi).
fn parse(text: &str) -> Vec<Case> {
    let mut cases = Vec::new();
    let mut lines = text.lines().filter(|l| !l.trim().is_empty() && !l.starts_with('#'));
    while let Some(head) = lines.next() {
        let w: Vec<&str> = head.split_whitespace().collect();
        assert!(w.len() == 22 && w[0] == "case", "bad header: {}", head);
        let field = |name: &str| field_of(&w, name);
        let n: usize = field("n").parse().unwrap();
        let mut rays = Vec::with_capacity(n);
        for _ in 0..n {
            let l = lines.next().expect("ray line");
            let v: Vec<f64> = l.split_whitespace().map(f64_from_hex).collect();
            assert!(v.len() == 8, "bad ray line: {}", l);
            rays.push(GeodesicState::new(v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7]));
        }
        cases.push(Case {
            key: w[1].to_string(),
            kind: field("kind").parse().unwrap(),
            spin: f64_from_hex(field("spin")),
            method: field("method").parse().unwrap(),
            tolerance: f64_from_hex(field("tol")),
            max_steps: field("max_steps").parse().unwrap(),
            step_size: f64_from_hex(field("step")),
            escape_radius: f64_from_hex(field("esc")),
            renormalize_interval: field("renorm").parse().unwrap(),
            initial_step: f64_from_hex(field("h0")),
            rays,
        });
    }
    cases
}

fn options(c: &Case, record_path: bool) -> IntegrationOptions {
    IntegrationOptions {
        method: match c.method {
            0 => IntegrationMethod::AdaptiveRKF45,
            1 => IntegrationMethod::RK4 { step_size: c.step_size },
            _ => IntegrationMethod::Symplectic { step_size: c.step_size },
        },
        tolerance: c.tolerance,
        initial_step: c.initial_step,
        max_steps: c.max_steps,
        escape_radius: c.escape_radius,
        renormalize_interval: c.renormalize_interval,
        record_path,
    }
}

fn state_json(s: &GeodesicState) -> String {
    let w: Vec<String> = s.x.iter().chain(s.p.iter()).map(|v| hex(*v)).collect();
    format!("[{}]", w.join(","))
}

fn run_case<M: Metric>(c: &Case, metric: &M, out: &mut String) {
    let opt = options(c, false);
    let trajs: Vec<Trajectory> = c.rays.iter().map(|r| integrate(r, metric, &opt)).collect();
    let states: Vec<String> = trajs.iter().map(|t| state_json(&t.final_state)).collect();
    let steps: Vec<String> = trajs.iter().map(|t| t.steps_taken.to_string()).collect();
    let term: Vec<String> = trajs.iter().map(|t| (t.termination as u8).to_string()).collect();
    let drift: Vec<String> = trajs.iter().map(|t| hex(t.max_hamiltonian_drift)).collect();
    // Trajectory.path of the first ray (record_path = true): point 0 is the state as passed in
    let path0: Vec<String> = match c.rays.first() {
        Some(r) => integrate(r, metric, &options(c, true)).path.unwrap_or_default().iter().map(state_json).collect(),
        None => Vec::new(),
    };
    write!(
        out,
        "\"{}\":{{\"out\":[{}],\"steps\":[{}],\"term\":[{}],\"drift\":[{}],\"path0\":[{}]}}",
        c.key,
        states.join(","),
        steps.join(","),
        term.join(","),
        drift.join(","),
        path0.join(",")
    )
    .unwrap();
}

fn main() {
    let args: Vec<String> = env::args().collect();
    if args.len() != 3 {
        eprintln!("usage: ref_vectors <inputs.txt> <ref_rays_v1.json>");
        std::process::exit(2);
    }
    let cases = parse(&fs::read_to_string(&args[1]).expect("read inputs"));
    let mut out = String::new();
    out.push_str("{\"format\":\"ref_rays_v1\",\"generator\":\"gravitas-core integrate() via tools/ref_vectors\",");
    write!(out, "\"target\":\"{}-{}\",\"cases\":{{", env::consts::ARCH, env::consts::OS).unwrap();
    for (i, c) in cases.iter().enumerate() {
        if i > 0 {
            out.push(',');
        }
        match c.kind {
            0 => run_case(c, &Kerr::new(1.0, c.spin), &mut out),
            1 => run_case(c, &Kerr::kerr_schild(1.0, c.spin), &mut out),
            _ => run_case(c, &Schwarzschild::new(1.0), &mut out),
        }
        eprintln!("{} ({} rays)", c.key, c.rays.len());
    }
    out.push_str("}}\n");
    fs::write(&args[2], out).expect("write output");
}
