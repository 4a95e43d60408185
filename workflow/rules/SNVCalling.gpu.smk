### MI355X replacement of LongSom's workflow/rules/SNVCalling.smk (same rule names' OUTPUT files).
#
# One fused rule produces every file the reference chain
#   SplitBam -> BaseCellCounter (x2) -> MergeCounts -> BaseCellCalling_step1 -> _step2 -> _step3
# leaves behind (except the per-cell-type BAMs, which only BaseCellCounter consumed; rule SplitBam_gpu
# below writes them on request, e.g. for the fusion branch).  `include: "rules/SNVCalling.gpu.smk"`
# instead of "rules/SNVCalling.smk" in workflow/Snakefile; config keys are unchanged.

GPU_SCRIPTS = str(workflow.basedir) + "/scripts_gpu"

rule SNVCalling_gpu:
    input:
        bam=f"{INPUT}/bam/{{id}}.bam",
        bai=f"{INPUT}/bam/{{id}}.bam.bai",
        barcodes="CellTypeReannotation/ReannotatedCellTypes/{id}.tsv" if REANNO else "Barcodes/{id}.tsv",
        ref=str(workflow.basedir)+config['Reference']['genome'],
        pon_LR="PoN/PoN/PoN_LR.tsv" if PON else [],
        pon_SR=str(workflow.basedir)+config['Reference']['PoN_SR'],
        RNA_editing=str(workflow.basedir)+config['Reference']['RNA_editing'],
    output:
        report="SNVCalling/SplitBam/{id}.report.txt",
        counts=expand("SNVCalling/BaseCellCounter/{{id}}/{{id}}.{celltype}.tsv", celltype=['Cancer','Non-Cancer']),
        merged="SNVCalling/MergeCounts/{id}.BaseCellCounts.AllCellTypes.tsv",
        step1="SNVCalling/BaseCellCalling/{id}.calling.step1.tsv",
        step2="SNVCalling/BaseCellCalling/{id}.calling.step2.tsv",
        step3="SNVCalling/BaseCellCalling/{id}.calling.step3.tsv",
    params:
        script=GPU_SCRIPTS+"/SNVCalling/longsom_gpu_snv.py",
        c=config['SNVCalling']['BaseCellCalling'],
        mapq=config['SNVCalling']['BaseCellCounter']['min_mapping_quality'],
    resources:
        gpu=1
    log:
        "logs/SNVCalling_gpu/{id}.log",
    benchmark:
        "benchmarks/SNVCalling_gpu/{id}.benchmark.txt"
    shell:
        r"""
        python {params.script} \
        --bam {input.bam} --meta {input.barcodes} --ref {input.ref} --id {wildcards.id} --outdir SNVCalling \
        --editing {input.RNA_editing} --pon_SR {input.pon_SR} --pon_LR {input.pon_LR} \
        --min_mapping_quality {params.mapq} \
        --min_cell_types {params.c[Min_cell_types]} --min_distance {params.c[min_distance]} \
        --max_gnomad_vaf {params.c[max_gnomAD_VAF]} --delta_vaf {params.c[deltaVAF]} --delta_mcf {params.c[deltaMCF]} \
        --min_ac_reads {params.c[min_ac_reads]} --min_ac_cells {params.c[min_ac_cells]} --clust_dist {params.c[clust_dist]} \
        --alpha1 {params.c[alpha1]} --beta1 {params.c[beta1]} --alpha2 {params.c[alpha2]} --beta2 {params.c[beta2]} > {log}
        """

rule SplitBam_gpu:
    input:
        bam=f"{INPUT}/bam/{{id}}.bam",
        barcodes="CellTypeReannotation/ReannotatedCellTypes/{id}.tsv" if REANNO else "Barcodes/{id}.tsv",
    output:
        expand("SNVCalling/SplitBam/{{id}}.{celltype}.bam", celltype=['Cancer','Non-Cancer'])
    params:
        script=GPU_SCRIPTS+"/PreProcessing/SplitBamCellTypes.py",
        mapq=config['SNVCalling']['BaseCellCounter']['min_mapping_quality'],
    shell:
        "python {params.script} --bam {input.bam} --meta {input.barcodes} --id {wildcards.id} --outdir SNVCalling/SplitBam --min_MQ {params.mapq}"
